"""Does the relative placement of the tensors a streaming kernel walks in lockstep matter (HBM channel aliasing)?
c = a + b over three 268 MB bf16 tensors carved out of one arena at different relative offsets."""
import torch
n = 128 * 128 * 128 * 64                      # elements of a level-1 activation tensor (bf16: 268 MB)
nb = n * 2
arena = torch.empty(4 * nb + (64 << 20), dtype=torch.uint8, device='cuda')
base = (arena.data_ptr() + (2 << 20) - 1) // (2 << 20) * (2 << 20) - arena.data_ptr()      # 2 MB aligned start
def view(off):
    return arena[off:off + nb].view(torch.bfloat16)
def timeit(fn, it=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
stride0 = (nb + (2 << 20) - 1) // (2 << 20) * (2 << 20)
for name, stag in [('2 MB aligned, back to back', 0), ('+256 B', 256), ('+1 KB', 1024), ('+4 KB', 4096), ('+16 KB', 16384), ('+64 KB', 65536),
                   ('+256 KB', 262144), ('+1 MB', 1 << 20), ('+4 KB*odd', 4096 * 3 + 256)]:
    a, b, c = view(base), view(base + stride0 + stag), view(base + 2 * (stride0 + stag))
    t = timeit(lambda: torch.add(a, b, out=c))
    print(f'{name:28s}: {t:7.1f} us  {3 * nb / t / 1e6:5.2f} TB/s')
