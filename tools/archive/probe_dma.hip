// Pins the LDS-DMA semantics wgrad6 relies on (gfx950): `buffer_load_dwordx4 v, s[0:3], 0 offen lds` writes lane l's 16 bytes
// to LDS byte (M0 + 16 l) whatever the lane's source offset is, and a lane whose offset lies beyond the descriptor's
// num_records gets ZEROS written (not skipped).   hipcc --offload-arch=gfx950 -O2 tools/archive/probe_dma.hip -o tools/probe_dma && tools/probe_dma
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
__device__ __forceinline__ void lds_dma16(u32x4_t rsrc, unsigned lds_dst, unsigned voff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst) : "memory");
}
__global__ void probe(const unsigned char* src, uint32_t* out, int mode) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    for (int i = lane; i < 2048 / 4; i += 64) reinterpret_cast<uint32_t*>(smem)[i] = 0xABABABABu;   // poison
    __syncthreads();
    const unsigned long long p = reinterpret_cast<unsigned long long>(src + (mode == 2 ? 4096 : 0));
    u32x4_t rs = {(unsigned)p, (unsigned)(p >> 32) & 0xffffu, 0x40000000u, 0x00020000u};
    unsigned voff;
    if (mode == 0) voff = ((lane * 37) % 64) * 16;                       // permuted sources
    else if (mode == 1) voff = (lane & 1) ? 0x80000000u : lane * 16;     // odd lanes out of range
    else voff = (lane < 32) ? (unsigned)(-1024 + lane * 16) : lane * 16; // base re-pointed past the start: negative offsets wrap out of range
    lds_dma16(rs, 512 /* LDS byte 512 */, voff);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 2048 / 4; i += 64) out[i] = reinterpret_cast<uint32_t*>(smem)[i];
}
int main() {
    std::vector<uint32_t> h(8192 / 4);
    for (size_t i = 0; i < h.size(); i++) h[i] = 0x10000000u + (uint32_t)i;
    unsigned char* d; uint32_t* o;
    hipMalloc(&d, 8192); hipMalloc(&o, 2048);
    hipMemcpy(d, h.data(), 8192, hipMemcpyHostToDevice);
    int bad = 0;
    for (int mode = 0; mode < 3; mode++) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 2048, 0, d, o, mode);
        std::vector<uint32_t> r(512);
        hipMemcpy(r.data(), o, 2048, hipMemcpyDeviceToHost);
        int nbad = 0;
        for (int i = 0; i < 512; i++) {
            uint32_t want = 0xABABABABu;
            const int b = i * 4 - 512;
            if (b >= 0 && b < 1024) {
                const int lane = b / 16, w = (b % 16) / 4;
                long off;
                if (mode == 0) off = ((lane * 37) % 64) * 16;
                else if (mode == 1) off = (lane & 1) ? -1 : lane * 16;
                else off = (lane < 32) ? -1 : 4096 + lane * 16;
                want = off < 0 ? 0u : h[off / 4 + w];
            }
            if (r[i] != want) { if (nbad < 4) printf("mode %d dword %d: got %08x want %08x\n", mode, i, r[i], want); nbad++; }
        }
        printf("mode %d: %s (%d mismatches)\n", mode, nbad ? "FAIL" : "ok", nbad);
        bad += nbad;
    }
    printf(bad ? "PROBE_DMA FAIL\n" : "PROBE_DMA OK\n");
    return bad ? 1 : 0;
}
