"""Host-side schedule of the bi-date Siamese U-Net on the HIP kernels.

This is the layer between the drop-in module surface (fabric_amd/models) and the
C ABI (include/bidate_hip.h): it owns the NHWC workspaces (torch tensors = caller
owned device memory), the packed-weight cache, and the order in which the fused
stages are enqueued on the current HIP stream.  It mirrors the data flow of the
reference's BiDateNet.forward (models/bidate_model.py:22-40) and of autograd through
it, with three structural differences chosen for the hardware:

  * both dates go through the shared encoder as ONE batch of 2B images with two
    BatchNorm statistic groups (the reference calls the encoder twice);
  * BatchNorm+ReLU is never materialised for 3x3-conv consumers: the consumer
    applies relu(z*scale+shift) while staging its input tile;
  * torch.cat of the skip and the upsampled map is never materialised: the decoder
    convolutions walk two source tensors along K.
"""
from dataclasses import dataclass, field

import torch

from . import _lib
from ._lib import BDN_BF16, BDN_BF16X2, BDN_BF16X3, BDN_F32, IN_BNRELU, IN_PLAIN, WG_ROLE, call, ptr, wg_flags

ENC_CH = (64, 128, 256, 512, 512)           # models/bidate_model.py:10-14
DEC_OUT = (256, 128, 64, 64)                # models/bidate_model.py:16-19
BN_EPS = 1e-5
BN_MOMENTUM = 0.1
FUSE_UPS_BS = 1          # tools/archive/ab_attr.py switch: upsample2x_bwd leaves the BatchNorm-backward partial sums (0: separate reduction pass)


def _round_up(v, m):
    return (v + m - 1) // m * m


@dataclass
class ConvLayer:
    name: str            # short id: e1a, e1b, ..., d4b
    conv: str            # state-dict prefix of the nn.Conv2d
    bn: str              # state-dict prefix of the nn.BatchNorm2d that follows
    cin_real: int
    cin: int             # padded to the kernel's channel granule
    cout: int
    level: int           # spatial level 1..5
    enc: bool


def build_layers(n_channels):
    """The 18 3x3 convolutions in forward order with their reference state-dict keys."""
    layers = []
    cp = _round_up(n_channels, 16)
    prev_real, prev = n_channels, cp
    for k in range(1, 6):
        base = 'inc.conv.conv' if k == 1 else f'down{k - 1}.mpconv.1.conv'
        co = ENC_CH[k - 1]
        layers.append(ConvLayer(f'e{k}a', f'{base}.0', f'{base}.1', prev_real, prev, co, k, True))
        layers.append(ConvLayer(f'e{k}b', f'{base}.3', f'{base}.4', co, co, co, k, True))
        prev_real = prev = co
    cprev = ENC_CH[4]
    for j in range(1, 5):
        k = 5 - j
        base = f'up{j}.conv.conv'
        ci = ENC_CH[k - 1] + cprev
        co = DEC_OUT[j - 1]
        layers.append(ConvLayer(f'd{j}a', f'{base}.0', f'{base}.1', ci, ci, co, k, False))
        layers.append(ConvLayer(f'd{j}b', f'{base}.3', f'{base}.4', co, co, co, k, False))
        cprev = co
    return layers


def param_order(n_channels):
    """State-dict keys of all learnable tensors in the order their gradients complete during
    backward (decoder top first, `inc` last); conv biases that feed a BatchNorm (analytically zero
    gradient) go last.  Used to lay out the flat gradient / parameter buffers so that gradient
    all-reduce buckets are contiguous slices that become ready front to back."""
    layers = build_layers(n_channels)
    order = ['outc.conv.weight', 'outc.conv.bias']
    for L in reversed(layers):
        order += [f'{L.bn}.weight', f'{L.bn}.bias', f'{L.conv}.weight']
    order += [f'{L.conv}.bias' for L in reversed(layers)]
    return order


class _LazyBufs(dict):
    """key -> tensor, allocated on first access.  The skips f_k, pooled maps and upsampled maps are not touched by a bf16x3 TRAINING
    forward (their producers store split GEMM operands instead), so a workspace that only ever trains in that setting never pays for them."""

    def __init__(self, shapes, alloc):
        super().__init__()
        self._shapes, self._alloc = shapes, alloc

    def __missing__(self, key):
        t = self[key] = self._alloc(*self._shapes[key])
        return t


class Workspace:
    """All device buffers for one (B, H, W) problem shape."""

    def __init__(self, eng, B, H, W, device):
        self.B, self.H, self.W = B, H, W
        td = eng.tdtype
        self.dims = []
        h, w = H, W
        for k in range(5):
            self.dims.append((h, w))
            h, w = h // 2, w // 2
        if min(self.dims[4]) < 1:
            raise RuntimeError(f'BiDateNet needs H,W >= 16 (got {H}x{W}): four 2x poolings')
        e = lambda *s: torch.empty(*s, dtype=td, device=device)
        f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=device)
        self.x0 = e(2 * B, H, W, eng.cp)
        self.z, self.bn = {}, {}
        pool_s, f_s, U_s = {}, {}, {}
        for L in eng.layers:
            hk, wk = self.dims[L.level - 1]
            n = 2 * B if L.enc else B
            self.z[L.name] = e(n, hk, wk, L.cout)
            self.bn[L.name] = f32(2 if L.enc else 1, 4, L.cout)
        for k in range(1, 6):
            hk, wk = self.dims[k - 1]
            if k >= 2:
                pool_s[k] = (2 * B, hk, wk, ENC_CH[k - 2])
            f_s[k] = (B, hk, wk, ENC_CH[k - 1])
        cprev = ENC_CH[4]
        for j in range(1, 5):
            hk, wk = self.dims[4 - j]
            U_s[j] = (B, hk, wk, cprev)
            cprev = DEC_OUT[j - 1]
        self.pool, self.f, self.U = _LazyBufs(pool_s, e), _LazyBufs(f_s, e), _LazyBufs(U_s, e)
        if not eng.x3:                                    # every other setting uses all of them in every forward: allocate now, in one place
            for d_ in (self.pool, self.f, self.U):
                for key in d_._shapes:
                    d_[key]
        lib = _lib.load()
        n_stats = n_bnb = n_wg = 1
        for L in eng.layers:
            hk, wk = self.dims[L.level - 1]
            n, ipg = (2 * B, B) if L.enc else (B, B)
            n_stats = max(n_stats, eng.mtiles(n, hk, wk, L.cin, L.cout, ipg) * 2 * L.cout)
            if L.name != 'e1a':                        # data-gradient launches with fused BatchNorm-backward sums: rows of THEIR tile plan
                n_stats = max(n_stats, eng.mtiles(n, hk, wk, L.cout, L.cin, ipg) * 2 * L.cin)
            if L.name == 'd4b':
                n_stats = max(n_stats, lib.bdn_outc_bwd_rows(eng.dt, B, hk, wk, L.cout) * 2 * L.cout)
            if not L.enc and L.name[2] == 'b' and L.level > 1:      # upsample2x_bwd_bs leaves this layer's BatchNorm-backward partials here
                n_stats = max(n_stats, lib.bdn_upsample2x_bwd_rows(eng.dt, B, hk, wk, L.cout) * 2 * L.cout)
            if L.enc:                                 # enc_skip_bwd leaves its BatchNorm-backward partials here too
                n_stats = max(n_stats, 2 * lib.bdn_enc_skip_bwd_rows(eng.dt, B, hk, wk, L.cout) * 2 * L.cout)
            n_bnb = max(n_bnb, lib.bdn_bn_bwd_workspace_bytes(eng.dt, n, hk, wk, L.cout, ipg) // 4)
            if not eng.x3:                            # bf16x3 sizes its doubled-operand workspace per call (split_buf)
                for blocks in (0, 512):               # 512: head room for A/B runs of engine.wgrad_blocks (any grid up to 512 fits)
                    n_wg = max(n_wg, lib.bdn_wgrad_workspace_bytes_ex(eng.dt, n, hk, wk, L.cout, L.cin, 0, ipg, IN_PLAIN, wg_flags(3, 0, blocks)) // 4)
        self.stats = f32(n_stats)
        self.bnws = torch.empty(2 * 64 * 2 * 1024, dtype=torch.float64, device=device)
        self._chain2 = None        # (stats, bnws) of the second forward chain, allocated when the two-chain forward first runs
        self.n_bnb, self.n_wg = n_bnb, n_wg
        L1 = eng.layers[0]                                # the first conv's fused weight-gradient GEMM runs beside another layer's
        self.n_wg1 = max(lib.bdn_wgrad_workspace_bytes_ex(dt_, 2 * B, H, W, L1.cout, L1.cin, 0, B, IN_PLAIN, wg_flags(3, 0, blocks)) // 4
                         for blocks in (0, 256, 512) for dt_ in ((eng.dt, BDN_BF16) if eng.x3 else (eng.dt,)))      # (0 / 512: the plain GEMM when it runs on the chain's stream, engine.last_wgrad_on_chain)
        self._bwd = None
        self._split = {}
        self._outc_ws = None
        self.logits = None
        self.x0_split = False      # bf16x3: the last pack_input stored the first convolution's split operand instead of x0
        self.leased = False        # True while a live autograd graph still needs this workspace's z / bn tables for its backward
        self.generation = 0        # bumped by every forward that overwrites the buffers (models/bidate_model.py checks it before a backward)

    def split_buf(self, which, numel):
        """bf16x3: buffer of a split GEMM operand ([.., 2C] bf16 = hi | lo), grown on demand.  Keys: ('a', layer) the layer's input
        operand, written by its training forward and read again by its weight-gradient GEMM; ('d', layer) its dz, split once on the
        chain's stream for the data-gradient conv and the weight-gradient GEMM (per layer: the weight-gradient stream may lag a
        layer behind); 'a' the shared operand buffer of eval forwards; 'p' the weight-gradient GEMM's workspace on its stream."""
        t = self._split.get(which)
        if t is None or t.numel() < numel:
            if t is not None:
                # the buffer being replaced may still be read by a weight-gradient GEMM queued on the second stream: its block must not be
                # handed out again before that stream has passed this point
                from . import streams
                t.record_stream(streams.get('wgrad', self.x0.device))
            t = torch.empty(numel, dtype=torch.bfloat16, device=self.x0.device)
            self._split[which] = t
        return t[:numel]

    def release_split(self):
        """bf16x3: drop the per-layer operand-split buffers (at B=16, 128x128 about 1.5 GB per workspace that has trained).  They are
        re-grown on demand by the next training forward; an eval-only phase after training calls this (BiDateNet.eval() does)."""
        self._split = {}

    def chain2(self):
        """Per-tile statistics buffer and finalize scratch of the forward's second chain (date 2 on its own stream): the two chains'
        convolutions are in flight at once, so they cannot share ws.stats / ws.bnws."""
        if self._chain2 is None:
            self._chain2 = (torch.empty_like(self.stats), torch.empty_like(self.bnws))
        return self._chain2

    def outc_ws(self, eng):
        """Scratch of bdn_outc_bwd (per-block partial classifier gradients)."""
        if self._outc_ws is None:
            n = _lib.load().bdn_outc_bwd_workspace_bytes(eng.dt, self.B, self.H, self.W, eng.layers[-1].cout, eng.n_classes)
            self._outc_ws = torch.empty(max(n // 4, 1), dtype=torch.float32, device=self.x0.device)
        return self._outc_ws

    def bwd_scratch(self, device):
        if self._bwd is None:
            self._bwd = dict(bnb=torch.empty(self.n_bnb, dtype=torch.float32, device=device),
                             wg=torch.empty(self.n_wg, dtype=torch.float32, device=device),
                             wg1=torch.empty(self.n_wg1, dtype=torch.float32, device=device),
                             sums=torch.empty(2 * 2 * 1024, dtype=torch.float32, device=device))
        return self._bwd


class BiDateEngine:
    """Enqueues forward / backward of BiDateNet(n_channels, n_classes) on the HIP library.

    precision: 'bf16' (bf16 activations and packed weights, fp32 accumulate -- throughput setting), 'fp32' (f32 storage +
    f32 MFMA -- the exact parity setting, 1/16 of the bf16 matrix rate) or 'bf16x3' (f32 storage; every GEMM operand split into
    bf16 hi + lo, three bf16 MFMAs per product, fp32 accumulate -- logits within 1e-3 of the reference at matrix-core speed);
    'bf16x3-fast' is bf16x3 with two-term backward GEMMs (same forward, gradients 2-5e-3 relative L2 away).
    Same kernels: one template parameter, resp. a three times longer reduction for the bf16 kernels."""

    def __init__(self, n_channels, n_classes, precision='bf16'):
        if precision not in ('bf16', 'fp32', 'bf16x3', 'bf16x3-fast'):
            raise ValueError(f"precision must be 'bf16', 'bf16x3', 'bf16x3-fast' or 'fp32', got {precision!r}")
        self.n_channels, self.n_classes = n_channels, n_classes
        self.precision = precision
        self.dt = BDN_BF16 if precision == 'bf16' else BDN_F32                  # storage type: what the HBM-bound kernels see
        self.x3 = precision in ('bf16x3', 'bf16x3-fast')
        self.mdt = BDN_BF16X3 if self.x3 else self.dt                           # what the GEMM kernels (conv3x3, wgrad, weight packing) see
        self.tdtype = torch.bfloat16 if precision == 'bf16' else torch.float32
        self.esize = 2 if precision == 'bf16' else 4
        self.cp = _round_up(n_channels, 16)
        self.layers = build_layers(n_channels)
        self._ws = {}
        self._packed = {}          # conv prefix -> (wf, wd) packed GEMM images
        self._packed_valid = False
        self._pack_desc = None
        self._packed_versions = None
        self._ev = None            # eval-mode tables: (parameter pointers, device descriptor, {layer: (scale, shift)}, identity BatchNorm table)
        # eval_fused: model.eval() forwards in the bf16 / fp32 settings run the eval-shaped schedule (_forward_eval: one launch per
        # conv -> BatchNorm -> ReLU stage, date product / pooling / classifier in the epilogues); False = the training kernels on a
        # running-statistics table (the round 1-5 path, kept for A/B and as the checker of the new one in tests)
        self.eval_fused = True
        # eval_pair: the second convolution of an encoder level runs on date-paired tiles (both dates of a pixel in one block: the skip product and
        # both pooled maps leave from LDS, neither activation is stored); False = one launch per date (date 2 reads date 1's stored activation)
        self.eval_pair = (1, 2, 3, 4, 5)       # encoder levels that take the paired form
        # The only tuning attributes (tools/archive/ab_flag.py A/Bs them in one process).  Everything round 1 and 2 measured and lost -- the
        # unfused BatchNorm-backward paths, relu(bn(z)) materialised for the weight gradient, the two-pass encoder skip backward,
        # release schedules of the weight-gradient GEMMs -- is gone from the product (DESIGN.md section 4 keeps the findings,
        # tools/experimental/ the code).
        # layers (64 output channels) whose BatchNorm backward is applied inside their data-gradient conv (bdn_conv3x3_dgrad_bb: dz = a g + b z + c
        # formed while the operand is staged) instead of by the bn_bwd_apply pass.  In-process A/B (tools/ab_cfg.py, round 5, three-constant
        # form on the masked gradient every fused producer stores): none +0.2 %, e1b = reference, e1b+d4a -0.2 %, +d3a -0.1 %, +d3b +0.1 %,
        # all four +0.2 % -- everything within ~0.2 % of noise on that box; on a second box e1b alone is +0.6 % against e1b+d4a.
        # The two full-resolution layers are kept (134 + 268 MB of dz reads less)
        self.fold_bn_bwd = ('e1b', 'd4a')
        # forward schedule.  fwd_chains = 2: the two dates go through encoder levels 1..fwd_chain_levels as two B-image chains on two streams
        # (_encoder_two_chains; bit-identical, measured +0.1...+1.3 % step time in round 5: off)
        self.fwd_chains = 1
        self.fwd_chain_levels = 3
        self._fwd_handoffs = {}
        self.wgrad_kernel = 0           # per-call kernel override of the weight-gradient GEMM (0 = the library's choice, _lib.WG_*)
        # bf16x3 settings: terms of the split product in the BACKWARD GEMMs.  The forward always keeps three (logits within 1e-3 of the reference:
        # north_star's bar).  'bf16x3' (the parity setting) keeps three in the backward as well; 'bf16x3-fast' is the explicit opt-in to two
        # (BDN_BF16X2: the filter rounded to bf16 in the data gradient, dz in the weight gradient) -- the gradients move by 2-5e-3 relative L2
        # (1 - cosine <= 1.3e-5) against the three-term backward for a 15-17 % shorter step
        self.x3_bwd_terms = 2 if precision == 'bf16x3-fast' else 3
        # bf16x3: convolutions whose operand is ONE float32 tensor of 64...512 channels (the second convolution of every double_conv) read it
        # directly -- BatchNorm+ReLU and the hi / lo split inside the staging (bdn_conv3x3_x3src), the split operand left for the weight gradient
        # as a by-product -- instead of behind a bdn_split_pack pass.  Each launch alone at B = 64 (tools/bench_x3_conv.py, us: split pass +
        # convolution -> one launch):  e1b 619 -> 690  e2b 444 -> 442  e3b 373 -> 397  e4b 345 -> 374  e5b 97 -> 103  d1b 53 -> 53  d2b 64 -> 64
        # d3b 83 -> 70  d4b 326 -> 273: the split pass streams at 5.9 TB/s and the conversion is ~50 VALU instructions per 16-byte unit of an
        # issue-bound kernel, so alone the sum is a wash (2.40 vs 2.47 ms).  IN the step (tools/ab_cfg.py, three-term backward, one process):
        # none 14.875 ms, d3b+d4b only 14.863, every eligible layer 14.838 (-0.25 %) -- the passes it removes ran beside the weight-gradient
        # queue.  True = every eligible layer (no split pass left in a step), a tuple of layer names = only those (eval forwards then take it
        # for operands of <= 128 channels), False = none (rounds 3-5; the checker of the fused form in tests)
        self.x3_src_f32 = True
        # the first convolution's weight gradient (the last GEMM of a backward pass, where the bf16 path runs its fused first-layer kernel on the chain's
        # stream) is launched on the chain's stream instead of behind the previous layer's GEMM on the second queue (fp32 / bf16x3 settings)
        self.last_wgrad_on_chain = True
        # the first convolution's weight gradient with its BatchNorm backward applied on load (bdn_conv3x3_wgrad_bnbwd: bf16 since round 2,
        # bf16x3 / bf16x3-fast since round 6); False = bn_bwd_apply + the generic GEMM (the checker of the fused form in tests)
        self.first_wgrad_fused = True
        # bf16x3, three-term backward: grid of the LAST GEMM on the second queue (inc's second convolution: 1.3 ms at the end of the step, when the
        # chain has little left to run beside it).  In-process A/B: 128 (as the others) 14.507 ms, 160 / 192 / 256 / 384 / 512 / 768: -0.3 / -0.5 /
        # -0.6 / -0.7 / -0.9 / -0.8 %; two-term backward: +-0 (not applied there).  0 = as the others
        self.x3_tail_wgrad_blocks = 512
        self.wgrad_blocks = 0           # per-call target grid of the weight-gradient GEMM (0 = the library's default: half the CUs)
        self._handoffs = {}             # device index -> reusable device-local events, one per hand-off of a backward pass
        self._diag_skip_wgrad = False
        self._diag_skip_handoff = 0     # timing diagnostics only (results WRONG): 1 = the weight-gradient GEMMs are released without the event hand-off
        self._diag_skip_reduce = 0      # timing diagnostics only (results WRONG): 1 = the split-K reductions of the weight-gradient GEMMs are not launched
        self.prof_pick = None      # with prof_filter: index of the one matching launch per step that gets the event pair
        self._prof_seen = 0
        self.prof_filter = None    # tuple of kernel instantiation names: only their launches are timed (an event pair is a ~150 us pipeline bubble)
        self.prof = None           # list collecting (kernel name, algorithmic flops, start event, end event)
        _lib.load()                # fail loudly now if the HIP extension is missing

    # ------------------------------------------------------------------ per-launch timing (bench.py roofline)
    def conv_kernel_name(self, n, h, w, c0, c1, cout, ipg):
        """Symbol of the conv3x3_kernel instantiation bdn_conv3x3 dispatches to: asked from the library's own dispatcher."""
        return _lib.load().bdn_conv3x3_variant(self.mdt, n, h, w, c0 + c1 if self.x3 else c0, 0 if self.x3 else c1, cout, ipg).decode()

    def _timed_conv(self, n, h, w, c0, c1, cout, ipg, *args, fn='bdn_conv3x3', bb=False):
        name = None
        if self.prof is not None:                   # BatchNorm-backward-on-load and float32-source launches have their own dispatchers: ask them
            if bb:
                name = _lib.load().bdn_conv3x3_dgrad_bb_variant(n, h, w, cout, ipg).decode()
            elif fn == 'bdn_conv3x3_x3src':
                name = _lib.load().bdn_conv3x3_x3src_variant(args[0], n, h, w, c0, cout, ipg).decode()
            else:
                name = self.conv_kernel_name(n, h, w, c0, c1, cout, ipg)
        if self.prof is None or (self.prof_filter is not None and name not in self.prof_filter):
            call(fn, *args)
            return
        if self.prof_pick is not None:              # sparse sampling: bracket only the prof_pick-th matching launch of this step
            self._prof_seen += 1
            if self._prof_seen - 1 != self.prof_pick:
                call(fn, *args)
                return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        call(fn, *args)
        e1.record()
        self.prof.append((name,
                          2.0 * n * h * w * cout * 9 * (c0 + c1), e0, e1))

    # ------------------------------------------------------------------ helpers
    def mtiles(self, n, h, w, c0, cout, ipg):
        """Spatial tiles (= rows of per-tile partial sums) of a convolution launch with c0 operand channels in this numerics setting: the
        bf16x3 kernels with the fused split product have their own tile plan (no 16 x 16 tiles)."""
        return _lib.load().bdn_conv3x3_num_mtiles_ex(self.mdt, n, h, w, c0, cout, ipg)

    def _side_stream(self, device):
        """The process-wide weight-gradient stream of the device (fabric_amd/streams.py)."""
        from . import streams
        return streams.get('wgrad', device)

    def workspace(self, B, H, W, device, slot=0):
        """A workspace of this shape that no live autograd graph owns (models/bidate_model.py leases the one its forward
        filled until the graph dies): a second forward of the same shape before backward() -- two micro-batches summed into
        one loss, a validation forward between forward and backward -- gets its own buffers instead of overwriting the
        activations the first graph's backward will read.  The fused TrainStep never leases, so it keeps one workspace."""
        key = (B, H, W, str(device), slot)      # slot: independent forward streams of one shape (scene inference on two streams)
        pool = self._ws.setdefault(key, [])
        for ws in pool:
            if not ws.leased:
                return ws
        pool.append(Workspace(self, B, H, W, device))
        return pool[-1]

    def drop_workspaces(self, slot):
        """Forget every un-leased workspace of `slot` (scene inference's second lane): the tensors go back to the caching allocator."""
        for key in [k for k in self._ws if k[4] == slot]:
            self._ws[key] = [w for w in self._ws[key] if w.leased]
            if not self._ws[key]:
                del self._ws[key]

    def _weights(self, L, P, need_wd):
        """Packed GEMM images of layer L (forward image, data-gradient image).  All 18 layers are (re)packed by
        ONE launch whenever any master weight changed since the last pack."""
        if not self._packed_valid:
            self._pack_all(P)
        ent = self._packed[L.conv]
        return ent[0], ent[1]

    def _pack_all(self, P):
        import struct
        if self.x3:
            # split filter images, three times the reduction length: [w_hi | w_hi | w_lo] (csrc/x3.hip); all layers in one launch
            ptrs = tuple(P[f'{L.conv}.weight'].data_ptr() for L in self.layers)
            if self._pack_desc is None or self._pack_desc[0] != ptrs:
                dev = P[f'{self.layers[0].conv}.weight'].device
                self._packed, rec = {}, b''
                for L in self.layers:
                    wf = torch.empty(L.cout, 9, 3 * L.cin, dtype=torch.bfloat16, device=dev)
                    wd = torch.empty(L.cin, 9, 3 * L.cout, dtype=torch.bfloat16, device=dev) if L.name != 'e1a' else None
                    self._packed[L.conv] = (wf, wd)
                    rec += struct.pack('<QQQiiii', P[f'{L.conv}.weight'].data_ptr(), wf.data_ptr(),
                                       wd.data_ptr() if wd is not None else 0, L.cout, L.cin_real, L.cin, 0)
                self._pack_desc = (ptrs, torch.frombuffer(bytearray(rec), dtype=torch.uint8).to(dev))
            call('bdn_pack_weights_multi', BDN_BF16X3, ptr(self._pack_desc[1]), len(self.layers), _lib.stream_ptr())
            self._packed_versions = tuple(P[f'{L.conv}.weight']._version for L in self.layers)
            self._packed_valid = True
            return
        ptrs = tuple(P[f'{L.conv}.weight'].data_ptr() for L in self.layers)
        if self._pack_desc is None or self._pack_desc[0] != ptrs:
            dev = P[f'{self.layers[0].conv}.weight'].device
            rec = b''
            self._packed = {}
            for L in self.layers:
                wf = torch.empty(L.cout, 9, L.cin, dtype=self.tdtype, device=dev)
                wd = torch.empty(L.cin, 9, L.cout, dtype=self.tdtype, device=dev) if L.name != 'e1a' else None
                self._packed[L.conv] = (wf, wd)
                rec += struct.pack('<QQQiiii', P[f'{L.conv}.weight'].data_ptr(), wf.data_ptr(),
                                   wd.data_ptr() if wd is not None else 0, L.cout, L.cin_real, L.cin, 0)
            desc = torch.frombuffer(bytearray(rec), dtype=torch.uint8).to(dev)
            self._pack_desc = (ptrs, desc)
        call('bdn_pack_weights_multi', self.dt, ptr(self._pack_desc[1]), len(self.layers), _lib.stream_ptr())
        self._packed_versions = tuple(P[f'{L.conv}.weight']._version for L in self.layers)
        self._packed_valid = True

    def invalidate_weights(self):
        self._packed_valid = False

    def release_split_buffers(self):
        """Drop the bf16x3 operand-split buffers of every workspace nobody leases (see Workspace.release_split)."""
        for pool in self._ws.values():
            for ws in pool:
                if not ws.leased:
                    ws.release_split()

    def _conv(self, ws, L, P, in0, c0, in1, c1, in_mode, in_bn, n, ipg, training, st, reuse_eval_bn=False, presplit=False,
              date=None, before_finalize=None, after_finalize=None):
        """One conv3x3 + BatchNorm statistics stage.  date = 0 / 1: the launch covers ONE date's B images of an encoder layer (two-chain
        forward): its slice of z, its row of the BatchNorm table, and -- for date 1, whose launches run beside date 0's -- the second
        chain's statistics buffers.  before_finalize / after_finalize: callables around the finalize launch (the running statistics
        see date 0 then date 1, so date 1's finalize is ordered behind date 0's)."""
        hk, wk = ws.dims[L.level - 1]
        wf, _ = self._weights(L, P, False)
        z = ws.z[L.name]
        stats, bnws, bn = ws.stats, ws.bnws, ws.bn[L.name]
        if date is not None:
            z, bn = z[date * n:(date + 1) * n], bn[date:date + 1]
            if date == 1:
                stats, bnws = ws.chain2()
        xs = self.x3_src_f32
        if self.x3 and in1 is None and not presplit and c0 % 64 == 0 and c0 <= 512 and xs and \
                (not isinstance(xs, (tuple, list, set)) or L.name in xs or (not training and c0 <= 128)):
            # one float32 source of >= 64 channels (the second convolution of every double_conv): BatchNorm+ReLU and the hi / lo split are
            # applied inside the convolution's staging (bdn_conv3x3_x3src); in training the tile's own pixels of the split operand are
            # stored as a by-product for the layer's weight-gradient GEMM -- no bdn_split_pack launch
            sp = ws.split_buf(('a', L.name), n * hk * wk * 2 * c0) if training else None
            self._timed_conv(n, hk, wk, c0, 0, L.cout, ipg,
                             self.mdt, ptr(in0), c0, in_mode, ptr(in_bn), ipg, ptr(wf), ptr(P[f'{L.conv}.bias']), ptr(z),
                             ptr(stats) if training else None, ptr(sp), n, hk, wk, L.cout, st, fn='bdn_conv3x3_x3src')
        else:
            if self.x3:
                # the operand split does the cat and the BatchNorm+ReLU the f32 kernel would apply on load
                # training: one buffer per layer, kept for the layer's weight-gradient GEMM (the same operand: no second split in backward)
                sp = ws.split_buf(('a', L.name) if training else 'a', n * hk * wk * 2 * (c0 + c1))
                if not presplit:                         # presplit: the producers of the operand (product_pool / upsample2x) stored it split already
                    call('bdn_split_pack', ptr(in0), c0, ptr(in1), c1, in_mode, ptr(in_bn), ipg, ptr(sp), n, hk, wk, st)
                in0, c0, in1, c1, in_mode, in_bn = sp, c0 + c1, None, 0, IN_PLAIN, None
            self._timed_conv(n, hk, wk, c0, c1, L.cout, ipg,
                             self.mdt, ptr(in0), c0, ptr(in1), c1, in_mode, ptr(in_bn), ipg,
                             ptr(wf), ptr(P[f'{L.conv}.bias']), ptr(z), ptr(stats) if training else None,
                             n, hk, wk, L.cout, st)
        G = n // ipg
        if training:
            nt = self.mtiles(n, hk, wk, c0 + c1, L.cout, ipg)
            if before_finalize is not None:
                before_finalize()
            call('bdn_bn_finalize', ptr(stats), nt, G, L.cout, ipg * hk * wk,
                 ptr(P[f'{L.bn}.weight']), ptr(P[f'{L.bn}.bias']), BN_EPS, BN_MOMENTUM,
                 ptr(P[f'{L.bn}.running_mean']), ptr(P[f'{L.bn}.running_var']),
                 ptr(P[f'{L.bn}.num_batches_tracked']), ptr(bn), ptr(bnws), st)
            if after_finalize is not None:
                after_finalize()
        elif not reuse_eval_bn:
            call('bdn_bn_eval', ptr(P[f'{L.bn}.weight']), ptr(P[f'{L.bn}.bias']),
                 ptr(P[f'{L.bn}.running_mean']), ptr(P[f'{L.bn}.running_var']), BN_EPS, G, L.cout, ptr(bn), st)
        return z, bn

    # ------------------------------------------------------------------ forward
    def forward(self, x_d1, x_d2, P, training=True, class_map=False):
        """x_d1, x_d2: [B,C,H,W] float32 CUDA tensors (reference layout).  P: state-dict-keyed tensors.
        Returns (logits [B,n_classes,H,W] float32, workspace).  class_map=True (eval mode only): returns the uint8 [B,H,W] map
        torch.max(logits, 1)[1] (train.py:199) instead of the logits -- on the eval-shaped schedule it comes straight out of the last
        convolution's epilogue."""
        self._prof_seen = 0
        if not (x_d1.is_cuda and x_d2.is_cuda):
            raise RuntimeError('fabric_amd: BiDateNet runs only on a ROCm device (MI355X); '
                               'inputs must be CUDA/HIP tensors -- there is no CPU path')
        if x_d1.shape != x_d2.shape or x_d1.dim() != 4 or x_d1.shape[1] != self.n_channels:
            raise RuntimeError(f'expected two [B,{self.n_channels},H,W] tensors, got {tuple(x_d1.shape)} and {tuple(x_d2.shape)}')
        x_d1 = x_d1.contiguous().float()
        x_d2 = x_d2.contiguous().float()
        B, C, H, W = x_d1.shape
        ws = self.workspace(B, H, W, x_d1.device)
        ws.generation += 1
        ws.x0_split = self.x3 and not class_map
        if ws.x0_split:            # bf16x3: the packed input leaves as the first convolution's [hi | lo] operand (no float32 x0, no split pass)
            sp = ws.split_buf(('a', self.layers[0].name) if training else 'a', 2 * B * H * W * 2 * self.cp)
            call('bdn_pack_input', BDN_BF16X3, ptr(x_d1), ptr(x_d2), ptr(sp), B, C, H, W, self.cp, _lib.stream_ptr())
        else:
            call('bdn_pack_input', self.dt, ptr(x_d1), ptr(x_d2), ptr(ws.x0), B, C, H, W, self.cp, _lib.stream_ptr())
        if class_map:
            if training:
                raise RuntimeError('class_map=True is an eval-mode output')
            cd = torch.empty(B, H, W, dtype=torch.uint8, device=x_d1.device)
            if self._use_eval_schedule():
                self._forward_eval(ws, P, mask=cd)
            else:
                logits = self._forward_packed(ws, P, False)
                call('bdn_argmax', ptr(logits), ptr(cd), B, self.n_classes, H, W, _lib.stream_ptr())
            return cd, ws
        return self._forward_packed(ws, P, training), ws

    def forward_tiles(self, scene_d1, scene_d2, origins, P, patch_size, reuse_eval_bn=False, slot=0, scene_mask=None):
        """Eval-mode forward of the tiles at `origins` (device int32 [n,2] = (y0,x0)) of a scene whose two dates
        are resident as [C,H,W] float32 band planes (train.py:190-197 without the host-side patch stack).
        reuse_eval_bn: the BatchNorm tables of this workspace (eval-shaped schedule: of the engine) are already those of P's running statistics.
        scene_mask: uint8 [H,W] device tensor -- the class index of every pixel of these tiles is written straight into it
        (utils/inference.py:187-236 ownership rule) and no logits are returned.
        Returns (logits [n,n_classes,p,p] float32 or None, workspace)."""
        if not (scene_d1.is_cuda and scene_d2.is_cuda and origins.is_cuda):
            raise RuntimeError('fabric_amd: scene planes and tile origins must be CUDA/HIP tensors -- there is no CPU path')
        if scene_d1.shape != scene_d2.shape or scene_d1.dim() != 3 or scene_d1.shape[0] != self.n_channels:
            raise RuntimeError(f'expected two [{self.n_channels},H,W] scenes, got {tuple(scene_d1.shape)} and {tuple(scene_d2.shape)}')
        if scene_d1.dtype != torch.float32 or scene_d2.dtype != torch.float32 or origins.dtype != torch.int32:
            raise RuntimeError('scene planes must be float32 and origins int32')
        if not (scene_d1.is_contiguous() and scene_d2.is_contiguous() and origins.is_contiguous()):
            raise RuntimeError('scene planes and origins must be contiguous')
        C, H, W = scene_d1.shape
        n, p = origins.shape[0], patch_size
        ws = self.workspace(n, p, p, scene_d1.device, slot)
        ws.generation += 1
        ws.x0_split = False
        call('bdn_gather_tiles', self.dt, ptr(scene_d1), ptr(scene_d2), ptr(origins), ptr(ws.x0),
             n, C, H, W, p, self.cp, _lib.stream_ptr())
        if scene_mask is not None:
            if scene_mask.dtype != torch.uint8 or tuple(scene_mask.shape) != (H, W) or not scene_mask.is_contiguous() or not scene_mask.is_cuda:
                raise RuntimeError(f'scene_mask must be a contiguous uint8 [{H},{W}] device tensor')
            if self._use_eval_schedule():
                self._forward_eval(ws, P, reuse_tables=reuse_eval_bn, mask=scene_mask, origins=origins, scene_hw=(H, W))
            else:
                logits = self._forward_packed(ws, P, False, reuse_eval_bn)
                call('bdn_argmax_stitch', ptr(logits), ptr(origins), ptr(scene_mask), n, logits.shape[1], p, H, W, _lib.stream_ptr())
            return None, ws
        return self._forward_packed(ws, P, False, reuse_eval_bn), ws

    def _forward_packed(self, ws, P, training, reuse_eval_bn=False):
        """The network on the packed input already in ws.x0."""
        B, H, W = ws.B, ws.H, ws.W
        dev = ws.x0.device
        st = _lib.stream_ptr()
        _lib.PHASE = 'fwd'
        by = {L.name: L for L in self.layers}
        if self._packed_valid and (self._packed_versions != tuple(P[f'{L.conv}.weight']._version for L in self.layers) or
                                   self._pack_desc[0] != tuple(P[f'{L.conv}.weight'].data_ptr() for L in self.layers)):
            # an optimizer touched the master weights (version counters), or they were re-pointed at other storage.
            # In-place writes through `p.data` (p.data.copy_(ema), p.data.clamp_()) bump NEITHER: call
            # invalidate_weights() after such an update (BiDateNet.load_state_dict / _apply do it themselves).
            self._packed_valid = False
        if not training and self._use_eval_schedule():
            return self._forward_eval(ws, P, reuse_tables=reuse_eval_bn)
        rb = reuse_eval_bn and not training
        k_first = 1
        if self.fwd_chains == 2 and training and not self.x3:
            k_first = self._encoder_two_chains(ws, P, by, st)
        # ---- shared encoder on both dates (2B images, 2 statistic groups)
        for k in range(k_first, 6):
            hk, wk = ws.dims[k - 1]
            La, Lb = by[f'e{k}a'], by[f'e{k}b']
            pre = self.x3 and training                      # bf16x3 training: pooled maps, skips and upsampled maps are stored as split operands
            src = ws.x0 if k == 1 else (None if pre else ws.pool[k])   # pool[k] was written together with the skip of level k-1
            za, bna = self._conv(ws, La, P, src, La.cin, None, 0, IN_PLAIN, None, 2 * B, B, training, st, rb,
                                 presplit=(pre and k > 1) or (k == 1 and ws.x0_split))
            zb, bnb = self._conv(ws, Lb, P, za, Lb.cin, None, 0, IN_BNRELU, bna, 2 * B, B, training, st, rb)
            if k < 5 and pre:
                Ld, Ln = by[f'd{5 - k}a'], by[f'e{k + 1}a']
                hn, wn = ws.dims[k]
                call('bdn_product_pool_split', ptr(zb), ptr(bnb), ptr(ws.split_buf(('a', Ld.name), B * hk * wk * 2 * Ld.cin)), 2 * Ld.cin, Ld.cin,
                     ptr(ws.split_buf(('a', Ln.name), 2 * B * hn * wn * 2 * Ln.cin)), B, hk, wk, ENC_CH[k - 1], st)
            elif k < 5:                                     # skip f_k and the pooled input of level k+1 in one pass over z
                call('bdn_product_pool', self.dt, ptr(zb), ptr(bnb), ptr(ws.f[k]), ptr(ws.pool[k + 1]), B, hk, wk, ENC_CH[k - 1], st)
            else:
                call('bdn_fuse_product', self.dt, ptr(zb), ptr(bnb), ptr(ws.f[k]), B, hk, wk, ENC_CH[k - 1], st)
        # ---- decoder on the fused skips
        prev, prev_bn, prev_mode, cprev = ws.f[5], None, IN_PLAIN, ENC_CH[4]
        for j in range(1, 5):
            k = 5 - j
            hk, wk = ws.dims[k - 1]
            hs, wsrc = ws.dims[k]
            La, Lb = by[f'd{j}a'], by[f'd{j}b']
            pre = self.x3 and training
            if pre:
                call('bdn_upsample2x_split', ptr(prev), prev_mode, ptr(prev_bn), ptr(ws.split_buf(('a', La.name), B * hk * wk * 2 * La.cin)),
                     2 * La.cin, ENC_CH[k - 1], La.cin, B, hs, wsrc, hk, wk, cprev, st)
            else:
                call('bdn_upsample2x', self.dt, ptr(prev), prev_mode, ptr(prev_bn), ptr(ws.U[j]),
                     B, hs, wsrc, hk, wk, cprev, st)
            za, bna = self._conv(ws, La, P, None if pre else ws.f[k], ENC_CH[k - 1], None if pre else ws.U[j], cprev, IN_PLAIN, None, B, B,
                                 training, st, rb, presplit=pre)
            zb, bnb = self._conv(ws, Lb, P, za, Lb.cin, None, 0, IN_BNRELU, bna, B, B, training, st, rb)
            prev, prev_bn, prev_mode, cprev = zb, bnb, IN_BNRELU, Lb.cout
        logits = torch.empty(B, self.n_classes, H, W, dtype=torch.float32, device=dev)
        call('bdn_outc_fwd', self.dt, ptr(prev), ptr(prev_bn), ptr(P['outc.conv.weight']), ptr(P['outc.conv.bias']),
             ptr(logits), B, H, W, cprev, self.n_classes, st)
        return logits

    # ------------------------------------------------------------------ eval-shaped forward (round 6)
    def _use_eval_schedule(self):
        return self.eval_fused and not self.x3

    def eval_tables(self, P):
        """Eval-mode BatchNorm of all 18 layers folded with the conv biases (bdn_bn_eval_fold_multi: one launch on the current stream).
        Returns {layer name: (scale [Cout], shift [Cout])}; the buffers are cached per parameter storage and REWRITTEN by every call
        (running statistics move without a version bump)."""
        import struct
        keys = [(f'{L.bn}.weight', f'{L.bn}.bias', f'{L.bn}.running_mean', f'{L.bn}.running_var', f'{L.conv}.bias') for L in self.layers]
        ptrs = tuple(P[k].data_ptr() for ks in keys for k in ks)
        if self._ev is None or self._ev[0] != ptrs:
            dev = P[keys[0][0]].device
            tab = torch.empty(sum(2 * L.cout for L in self.layers), dtype=torch.float32, device=dev)
            rec, views, off = b'', {}, 0
            for L, ks in zip(self.layers, keys):
                out = tab[off:off + 2 * L.cout]
                off += 2 * L.cout
                rec += struct.pack('<QQQQQQii', *(P[k].data_ptr() for k in ks), out.data_ptr(), L.cout, 0)
                views[L.name] = (out[:L.cout], out[L.cout:])
            ident = torch.zeros(1, 4, self.layers[-1].cout, dtype=torch.float32, device=dev)   # {mean 0, invstd 1, scale 1, shift 0}: relu(bn(a)) = a for a >= 0
            ident[:, 1:3] = 1.0
            self._ev = (ptrs, torch.frombuffer(bytearray(rec), dtype=torch.uint8).to(dev), views, ident, tab)
        call('bdn_bn_eval_fold_multi', ptr(self._ev[1]), len(self.layers), max(L.cout for L in self.layers), BN_EPS, _lib.stream_ptr())
        return self._ev[2]

    def _forward_eval(self, ws, P, reuse_tables=False, mask=None, origins=None, scene_hw=None):
        """model.eval() forward on the packed input in ws.x0 (reference: train.py:125-172 validation, train.py:182-205 full-scene inference).
        Nothing depends on batch statistics, so every conv -> BatchNorm -> ReLU stage is ONE launch whose epilogue applies the folded
        running-statistics affine + ReLU and stores the ACTIVATION (bdn_conv3x3_eval): consumers stage plain bytes, there are no statistics
        partials, no finalize / bn_eval launches.  The second convolution of an encoder level runs on date-PAIRED tiles (both dates of the
        same pixels in one block, bdn_conv3x3_eval_pair): the skip relu(x_d2 * x_d1) (models/bidate_model.py:35-38) and both pooled maps
        leave from LDS and neither date's activation reaches HBM; product_pool / fuse_product launches are gone.  The last
        decoder convolution carries the 1x1 classifier (and, for scene inference, argmax + stitching) in its epilogue.
        Returns logits [B,n_classes,H,W] float32, or None when `mask` is given (uint8 [B,H,W], or the scene mask [Hs,Ws] with `origins`)."""
        B, H, W = ws.B, ws.H, ws.W
        dev = ws.x0.device
        st = _lib.stream_ptr()
        _lib.PHASE = 'fwd'
        by = {L.name: L for L in self.layers}
        if self._packed_valid and (self._packed_versions != tuple(P[f'{L.conv}.weight']._version for L in self.layers) or
                                   self._pack_desc[0] != tuple(P[f'{L.conv}.weight'].data_ptr() for L in self.layers)):
            self._packed_valid = False
        ev = self._ev[2] if (reuse_tables and self._ev is not None) else self.eval_tables(P)

        def stage(L, in0, c0, in1, c1, out, n, hk, wk, mul=None, pool=None):
            wf, _ = self._weights(L, P, False)
            sc, sh = ev[L.name]
            call('bdn_conv3x3_eval', self.dt, ptr(in0), c0, ptr(in1), c1, ptr(wf), ptr(sc), ptr(sh), ptr(out), ptr(mul), ptr(pool),
                 n, hk, wk, L.cout, st)

        # ---- shared encoder: the first conv of a level on both dates at once, the second per date
        for k in range(1, 6):
            hk, wk = ws.dims[k - 1]
            La, Lb = by[f'e{k}a'], by[f'e{k}b']
            src = ws.x0 if k == 1 else ws.pool[k]
            za, zb = ws.z[La.name], ws.z[Lb.name]
            stage(La, src, La.cin, None, 0, za, 2 * B, hk, wk)
            pn = ws.pool[k + 1] if k < 5 else None
            if k in self.eval_pair:
                wf, _ = self._weights(Lb, P, False)
                sc, sh = ev[Lb.name]
                call('bdn_conv3x3_eval_pair', self.dt, ptr(za), Lb.cin, ptr(wf), ptr(sc), ptr(sh), ptr(ws.f[k]), ptr(pn), B, hk, wk, Lb.cout, st)
            else:                                    # per date: date 1 stores its activation, date 2 multiplies its own with it in the copy-out
                stage(Lb, za[:B], Lb.cin, None, 0, zb[:B], B, hk, wk, pool=pn[:B] if k < 5 else None)
                stage(Lb, za[B:], Lb.cin, None, 0, ws.f[k], B, hk, wk, mul=zb[:B], pool=pn[B:] if k < 5 else None)
        # ---- decoder on the fused skips
        prev, cprev = ws.f[5], ENC_CH[4]
        logits = None
        for j in range(1, 5):
            k = 5 - j
            hk, wk = ws.dims[k - 1]
            hs, wsrc = ws.dims[k]
            La, Lb = by[f'd{j}a'], by[f'd{j}b']
            call('bdn_upsample2x', self.dt, ptr(prev), IN_PLAIN, None, ptr(ws.U[j]), B, hs, wsrc, hk, wk, cprev, st)
            stage(La, ws.f[k], ENC_CH[k - 1], ws.U[j], cprev, ws.z[La.name], B, hk, wk)
            if j < 4 or self.n_classes > 2:
                stage(Lb, ws.z[La.name], Lb.cin, None, 0, ws.z[Lb.name], B, hk, wk)
            else:
                wf, _ = self._weights(Lb, P, False)
                sc, sh = ev[Lb.name]
                if mask is None:
                    logits = torch.empty(B, self.n_classes, H, W, dtype=torch.float32, device=dev)
                call('bdn_conv3x3_eval_cls', self.dt, ptr(ws.z[La.name]), Lb.cin, ptr(wf), ptr(sc), ptr(sh), None,
                     ptr(P['outc.conv.weight']), ptr(P['outc.conv.bias']), self.n_classes, ptr(logits), ptr(mask), ptr(origins),
                     scene_hw[0] if scene_hw else 0, scene_hw[1] if scene_hw else 0, B, hk, wk, Lb.cout, st)
            prev, cprev = ws.z[Lb.name], Lb.cout
        if self.n_classes > 2:                       # wider heads: the stand-alone classifier on the stored activation (identity BatchNorm table)
            logits = torch.empty(B, self.n_classes, H, W, dtype=torch.float32, device=dev)
            call('bdn_outc_fwd', self.dt, ptr(prev), ptr(self._ev[3]), ptr(P['outc.conv.weight']), ptr(P['outc.conv.bias']),
                 ptr(logits), B, H, W, cprev, self.n_classes, st)
            if mask is not None:
                if origins is not None:
                    call('bdn_argmax_stitch', ptr(logits), ptr(origins), ptr(mask), B, self.n_classes, H, scene_hw[0], scene_hw[1], st)
                else:
                    call('bdn_argmax', ptr(logits), ptr(mask), B, self.n_classes, H, W, st)
                return None
        return logits

    def _fwd_handoff(self, dev, i):
        pool = self._fwd_handoffs.setdefault(dev.index, [])
        while len(pool) <= i:
            from .streams import HandOff
            with torch.cuda.device(dev):
                pool.append(HandOff())
        return pool[i]

    def _encoder_two_chains(self, ws, P, by, st):
        """Encoder levels 1..fwd_chain_levels with the two dates as two independent B-image chains: date 0 on the current (chain) stream,
        date 1 on the library's second stream (idle in forward).  The reference runs the dates one after the other through the same
        modules (models/bidate_model.py:23-33); here they were one 2B batch with two statistic groups -- the split changes no arithmetic
        (same tiles, same per-group reductions) but lets one date's convolutions run while the other's dependent reduce / finalize / pool
        launches drain.  Ordering kept by events: the running statistics and num_batches_tracked are updated by date 0's finalize first,
        then by date 1's (reference order).  Date 0's chain pools its own map; date 1's chain forms the skip f_k (it needs date 0's z and
        table: ordered behind date 0's finalize of that layer already) together with ITS pooled map in one pass.  Measured (round 5, six
        more variants of where the products run / which stream / how many levels): +0.1...+1.3 % step time -- the option stays OFF.
        Returns the first level the joined schedule continues with."""
        from . import streams
        B = ws.B
        dev = ws.x0.device
        main = torch.cuda.current_stream(dev)
        second = streams.get('wgrad', dev)
        Lmax = max(1, min(4, self.fwd_chain_levels))
        ev = [0]

        def new_ev():
            ev[0] += 1
            return self._fwd_handoff(dev, ev[0] - 1)

        self._weights(self.layers[0], P, False)              # (re)pack the filter images NOW, on the chain stream, in front of the fork
        start = new_ev()
        start.signal(main)                                   # packed input and packed weights are ready
        start.wait(second)
        chains = ((0, main, st), (1, second, second.cuda_stream))
        fin_events = {}
        for k in range(1, Lmax + 1):
            hk, wk = ws.dims[k - 1]
            La, Lb = by[f'e{k}a'], by[f'e{k}b']
            src = ws.x0 if k == 1 else ws.pool[k]
            for L in (La, Lb):
                fin_events[L.name] = new_ev()
            for d, stream, sp in chains:
                with torch.cuda.stream(stream):
                    def order(L, d=d, stream=stream):
                        if d == 0:
                            return dict(after_finalize=lambda: fin_events[L.name].signal(stream))
                        return dict(before_finalize=lambda: fin_events[L.name].wait(stream))
                    za, bna = self._conv(ws, La, P, src[d * B:(d + 1) * B], La.cin, None, 0, IN_PLAIN, None, B, B, True, sp, date=d, **order(La))
                    zb, bnb = self._conv(ws, Lb, P, za, Lb.cin, None, 0, IN_BNRELU, bna, B, B, True, sp, date=d, **order(Lb))
                    if d == 0:
                        call('bdn_bnrelu_pool', self.dt, ptr(zb), ptr(bnb), B, ptr(ws.pool[k + 1][:B]), B, hk, wk, ENC_CH[k - 1], sp)
                    else:
                        call('bdn_product_pool_dates', self.dt, ptr(ws.z[Lb.name]), ptr(ws.bn[Lb.name]), ptr(ws.f[k]), ptr(ws.pool[k + 1]), 2,
                             B, hk, wk, ENC_CH[k - 1], sp)
        join = new_ev()
        join.signal(second)
        join.wait(main)
        return Lmax + 1

    # ------------------------------------------------------------------ backward
    def backward(self, ws, dlogits, P, grads, on_ready=None, zero_bias_grads=True, wgrad_stream=True):
        """Gradient of the last training-mode forward on `ws`.

        dlogits: [B,n_classes,H,W] float32.  grads: dict key -> preallocated float32 tensor (reference
        parameter shapes) that is OVERWRITTEN.  on_ready(keys) is called after the kernels producing
        those gradients have been enqueued (used to launch gradient all-reduce buckets early).
        zero_bias_grads=False: the caller guarantees the conv-bias gradient tensors already hold zeros (nobody
        ever writes them), which saves 18 fill launches per step.
        wgrad_stream=True: the weight-gradient GEMMs (off the critical dz -> dgrad -> dz chain, MFMA-bound) are
        enqueued on a second HIP stream so they overlap the HBM-bound BatchNorm-backward / unpool / upsample
        kernels of the chain; the main stream joins it before returning."""
        B, H, W = ws.B, ws.H, ws.W
        dev = dlogits.device
        dlogits = dlogits.contiguous().float()
        st = _lib.stream_ptr()
        _lib.PHASE = 'bwd'
        by = {L.name: L for L in self.layers}
        sc = ws.bwd_scratch(dev)
        td, es = self.tdtype, self.esize
        e = lambda *s: torch.empty(*s, dtype=td, device=dev)
        ready = on_ready or (lambda keys: None)
        main = torch.cuda.current_stream(dev)
        side = self._side_stream(dev) if wgrad_stream else None

        def bn_bwd(L, dA, ldA, n, ipg, fused_rows=0):
            """BatchNorm+ReLU backward of layer L.  fused_rows > 0: the kernel that produced dA already left the
            per-tile partial sums (sum g, sum g*z) in ws.stats, fused_rows rows per statistic group."""
            hk, wk = ws.dims[L.level - 1]
            if fused_rows and self.x3:
                # bf16x3: dz leaves the pass as the [hi | lo] operand of its two consumers (per-layer buffer: the weight-gradient stream may
                # lag a layer behind); there is no float32 dz and no split pass over it
                sp = ws.split_buf(('d', L.name), n * hk * wk * 2 * L.cout)
                call('bdn_bn_bwd_apply_split', dA, ldA, ptr(ws.z[L.name]), ptr(ws.bn[L.name]), ipg, n, hk, wk, L.cout,
                     ptr(ws.stats), fused_rows, 1, ptr(sc['sums']), ptr(grads[f'{L.bn}.weight']),
                     ptr(grads[f'{L.bn}.bias']), ptr(sp), ptr(ws.bnws), st)
                return None
            dz = e(n, hk, wk, L.cout)
            if fused_rows:
                call('bdn_bn_bwd_apply', self.dt, dA, ldA, ptr(ws.z[L.name]), ptr(ws.bn[L.name]), ipg, n, hk, wk, L.cout,
                     ptr(ws.stats), fused_rows, 1, ptr(sc['sums']), ptr(grads[f'{L.bn}.weight']),
                     ptr(grads[f'{L.bn}.bias']), ptr(dz), ptr(ws.bnws), st)
            else:
                call('bdn_bn_bwd', self.dt, dA, ldA, ptr(ws.z[L.name]), ptr(ws.bn[L.name]), ipg, n, hk, wk, L.cout,
                     ptr(sc['bnb']), ptr(sc['sums']), ptr(grads[f'{L.bn}.weight']), ptr(grads[f'{L.bn}.bias']), ptr(dz), st)
            return dz

        def fold_dgrad(L, dA, n, ipg, fused_rows, prev=None):
            """BatchNorm+ReLU backward of layer L applied while its data-gradient conv stages dz (no bn_bwd_apply pass): the partial sums in
            ws.stats are finalized (sums, dgamma, dbeta), then ONE kernel forms dz on load, convolves it and stores it for the weight
            gradient.  Returns (dz, dA_prev[, rows])."""
            hk, wk = ws.dims[L.level - 1]
            G = n // ipg
            call('bdn_bn_bwd_finalize', ptr(ws.bn[L.name]), G, L.cout, ptr(ws.stats), fused_rows, 1, ptr(sc['sums']),
                 ptr(grads[f'{L.bn}.weight']), ptr(grads[f'{L.bn}.bias']), ptr(ws.bnws), st)
            _, wd = self._weights(L, P, True)
            dz, out = e(n, hk, wk, L.cout), e(n, hk, wk, L.cin)
            has = prev is not None
            self._timed_conv(n, hk, wk, L.cout, 0, L.cin, ipg,
                             self.mdt, dA, L.cout, ptr(ws.z[L.name]), ptr(ws.bn[L.name]), ptr(sc['sums']), ipg, ptr(wd), ptr(out),
                             ptr(ws.z[prev.name]) if has else None, ptr(ws.bn[prev.name]) if has else None, ptr(ws.stats) if has else None,
                             ptr(dz), n, hk, wk, L.cin, st, fn='bdn_conv3x3_dgrad_bb', bb=True)
            if has:
                return dz, out, self.mtiles(n, hk, wk, L.cout, L.cin, ipg) // G
            return dz, out

        def wgrad_call(L, dz, in0, c0, in1, c1, mode, in_bn, n, ipg, hk, wk, stp, part_key='p'):
            """The weight-gradient GEMM and its reduction; with profiling on, the GEMM alone sits between two events
            recorded on the stream it is launched on."""
            lib = _lib.load()
            if self.x3:
                # both operands were split already: the activations by this layer's forward, dz by split_dz() on the chain's stream
                # (per-layer buffers: the weight-gradient stream may still read one while the chain splits the next layer's)
                sd = ws.split_buf(('d', L.name), n * hk * wk * 2 * L.cout)
                sw = ws.split_buf(('a', L.name), n * hk * wk * 2 * (c0 + c1))
                blk_ = self.x3_tail_wgrad_blocks if (L.name == 'e1b' and self.x3_tail_wgrad_blocks and self.x3_bwd_terms == 3) else self.wgrad_blocks
                flg = wg_flags(1 if self._diag_skip_reduce else 3, 0, blk_)      # (_diag_skip_reduce: timing diagnostics only)
                xdt = BDN_BF16X2 if self.x3_bwd_terms == 2 else BDN_BF16X3
                nb = lib.bdn_wgrad_workspace_bytes_ex(xdt, n, hk, wk, L.cout, c0 + c1, 0, ipg, IN_PLAIN, flg)
                part = ws.split_buf(part_key, nb // 2)        # ('p1': the one GEMM that runs on the chain's stream beside the queue's own)
                call('bdn_conv3x3_wgrad_ex', xdt, ptr(sd), L.cout, ptr(sw), c0 + c1, None, 0, IN_PLAIN, None, ipg,
                     ptr(part), ptr(grads[f'{L.conv}.weight']), L.cin_real, n, hk, wk, flg, stp)
                return
            wk_, blk_ = self.wgrad_kernel, self.wgrad_blocks
            args = (self.dt, ptr(dz), L.cout, ptr(in0), c0, ptr(in1), c1, mode, ptr(in_bn), ipg,
                    ptr(sc['wg1' if part_key == 'p1' else 'wg']), ptr(grads[f'{L.conv}.weight']), L.cin_real, n, hk, wk)
            name = None
            if self.prof is not None:
                v = lib.bdn_conv3x3_wgrad_variant(self.dt, n, hk, wk, L.cout, c0, c1, ipg, mode, wg_flags(3, wk_, blk_))
                if v == WG_ROLE:
                    name = f'wgrad7_kernel<{"true" if mode == IN_BNRELU else "false"}>'
                else:
                    small = wk <= 8 and hk <= 8 and ipg % 2 == 0
                    name = (f'wgrad_kernel<{"bf16" if self.precision == "bf16" else "f32"},8,{"8,2" if small else "16,1"},'
                            f'{"true" if c0 + c1 <= 32 else "false"}>')
                if self.prof_filter is not None and name not in self.prof_filter:
                    name = None
                elif self.prof_pick is not None:
                    self._prof_seen += 1
                    if self._prof_seen - 1 != self.prof_pick:
                        name = None
            if name is None:
                call('bdn_conv3x3_wgrad_ex', *args, wg_flags(1 if self._diag_skip_reduce else 3, wk_, blk_), stp)
                return
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            call('bdn_conv3x3_wgrad_ex', *args, wg_flags(1, wk_, blk_), stp)
            e1.record()
            call('bdn_conv3x3_wgrad_ex', *args, wg_flags(2, wk_, blk_), stp)
            self.prof.append((name, 2.0 * n * hk * wk * L.cout * 9 * (c0 + c1), e0, e1))

        n_hand = [0]

        def handoff(src, dst):
            """Order what `dst` enqueues from now on behind what `src` has enqueued: a device-local event without the system-scope
            fence of a default event (streams.HandOff; 6.213 -> 6.187 ms per step in one process), one reusable event per hand-off."""
            pool = self._handoffs.setdefault(dev.index, [])      # events belong to the device they were created on
            if n_hand[0] == len(pool):
                from .streams import HandOff
                with torch.cuda.device(dev):
                    pool.append(HandOff())
            ho = pool[n_hand[0]]
            n_hand[0] += 1
            if self._diag_skip_handoff:
                return
            ho.signal(src)
            ho.wait(dst)

        def split_dz(L, dz, n, ipg):
            """bf16x3: the [hi | lo] split of layer L's dz, once, for its data-gradient conv AND its weight-gradient GEMM."""
            hk, wk = ws.dims[L.level - 1]
            sp = ws.split_buf(('d', L.name), n * hk * wk * 2 * L.cout)
            call('bdn_split_pack', ptr(dz), L.cout, None, 0, IN_PLAIN, None, ipg, ptr(sp), n, hk, wk, st)
            return sp

        def wgrad(L, dz, in0, c0, in1, c1, mode, in_bn, n, ipg, on_main=False):
            if self.x3 and dz is not None:
                split_dz(L, dz, n, ipg)              # on the chain's stream, before the hand-off below (bn_bwd already left the split otherwise)
            if self._diag_skip_wgrad:                # tools/ab_step.py diagnostic only: how long is the dz chain alone?
                return
            hk, wk = ws.dims[L.level - 1]
            keys = [f'{L.bn}.weight', f'{L.bn}.bias', f'{L.conv}.weight', f'{L.conv}.bias']
            if on_main and side is not None:
                # the LAST weight gradient of the pass (the first convolution's) on the chain's own stream: nothing of the chain is left to
                # run and the second queue is still busy with the layer before it -- the two GEMMs run side by side instead of one behind
                # the other (bf16x3: 0.36 ms behind a 1.3 ms GEMM at the end of the step).  The chain then joins the second queue: the
                # bucket this ready() may release holds gradients whose GEMMs are still queued there.
                wgrad_call(L, dz, in0, c0, in1, c1, mode, in_bn, n, ipg, hk, wk, st, part_key='p1')
                if zero_bias_grads:
                    grads[f'{L.conv}.bias'].zero_()
                handoff(side, main)
                ready(keys)
                return
            if side is None:
                wgrad_call(L, dz, in0, c0, in1, c1, mode, in_bn, n, ipg, hk, wk, st)
                if zero_bias_grads:                  # feeds a BatchNorm: gradient is identically zero
                    grads[f'{L.conv}.bias'].zero_()
                ready(keys)
                return
            handoff(main, side)                      # dz, the BatchNorm gradients and everything before them
            with torch.cuda.stream(side):
                wgrad_call(L, dz, in0, c0, in1, c1, mode, in_bn, n, ipg, hk, wk, side.cuda_stream)
                if zero_bias_grads:
                    grads[f'{L.conv}.bias'].zero_()
                ready(keys)                          # a bucket all-reduce launched here is ordered behind this wgrad

        def dgrad(L, dz, n, ipg, prev=None):
            """Data gradient of layer L's conv.  prev = the layer whose relu(bn(z)) is this conv's input: its
            BatchNorm-backward partial sums are then produced by the epilogue (returns rows per statistic group)."""
            hk, wk = ws.dims[L.level - 1]
            _, wd = self._weights(L, P, True)
            out = e(n, hk, wk, L.cin)
            if self.x3:
                dz = ws.split_buf(('d', L.name), n * hk * wk * 2 * L.cout)     # written by split_dz() when this layer's wgrad was released
            ddt = BDN_BF16X2 if (self.x3 and self.x3_bwd_terms == 2) else self.mdt
            if prev is None:
                self._timed_conv(n, hk, wk, L.cout, 0, L.cin, ipg,
                                 ddt, ptr(dz), L.cout, None, 0, IN_PLAIN, None, ipg,
                                 ptr(wd), None, ptr(out), None, n, hk, wk, L.cin, st)
                return out
            self._timed_conv(n, hk, wk, L.cout, 0, L.cin, ipg,
                             ddt, ptr(dz), L.cout, ptr(wd), ptr(out), ptr(ws.z[prev.name]), ptr(ws.bn[prev.name]),
                             ipg, ptr(ws.stats), n, hk, wk, L.cin, st, fn='bdn_conv3x3_dgrad_bs')
            rows = self.mtiles(n, hk, wk, L.cout, L.cin, ipg) // (n // ipg)
            return out, rows

        # ---- classifier: its data gradient is never stored -- bdn_outc_bwd leaves the BatchNorm-backward partial sums of d4b, and
        # d4b's BatchNorm backward recomputes dA = round(sum_k dlogits[k] w[k][c]) (bit-identical dz, 134 MB less footprint)
        L4b = by['d4b']
        call('bdn_outc_bwd', self.dt, ptr(dlogits), ptr(ws.z['d4b']), ptr(ws.bn['d4b']), ptr(P['outc.conv.weight']),
             None, ptr(grads['outc.conv.weight']), ptr(grads['outc.conv.bias']), ptr(ws.stats),
             ptr(ws.outc_ws(self)), B, H, W, L4b.cout, self.n_classes, st)
        rows_head = _lib.load().bdn_outc_bwd_rows(self.dt, B, H, W, L4b.cout)
        ready(['outc.conv.weight', 'outc.conv.bias'])
        # ---- decoder
        dA_ptr, ldA, rows_up = None, 0, 0
        fold = set(self.fold_bn_bwd) if (self.mdt == BDN_BF16) else set()
        keep = []
        dcat = {}
        dF5 = None
        for j in range(4, 0, -1):
            k = 5 - j
            hk, wk = ws.dims[k - 1]
            hs, wsrc = ws.dims[k]
            La, Lb = by[f'd{j}a'], by[f'd{j}b']
            ck = ENC_CH[k - 1]
            cprev = La.cin - ck
            folded_b = False
            if j == 4:
                # bf16x3: dz leaves the pass as the [hi | lo] operand of its two consumers (no float32 dz, no split pass over it)
                dzb = None if self.x3 else e(B, hk, wk, Lb.cout)
                dz_out = ws.split_buf(('d', Lb.name), B * hk * wk * 2 * Lb.cout) if self.x3 else dzb
                call('bdn_bn_bwd_finalize', ptr(ws.bn[Lb.name]), 1, Lb.cout, ptr(ws.stats), rows_head, 1, ptr(sc['sums']),
                     ptr(grads[f'{Lb.bn}.weight']), ptr(grads[f'{Lb.bn}.bias']), ptr(ws.bnws), st)
                call('bdn_outc_bn_bwd_apply', self.mdt if self.x3 else self.dt, ptr(dlogits), ptr(P['outc.conv.weight']), ptr(ws.z[Lb.name]),
                     ptr(ws.bn[Lb.name]), B, ptr(sc['sums']), ptr(dz_out), B, hk, wk, Lb.cout, self.n_classes, st)
            elif Lb.name in fold and rows_up and ldA == Lb.cout == 64 and min(hk, wk) > 8:
                dzb, dAa, rows = fold_dgrad(Lb, dA_ptr, B, B, rows_up, prev=La)
                folded_b = True
            else:
                dzb = bn_bwd(Lb, dA_ptr, ldA, B, B, fused_rows=rows_up)     # dA came from upsample2x_bwd(_bs)
            wgrad(Lb, dzb, ws.z[La.name], Lb.cin, None, 0, IN_BNRELU, ws.bn[La.name], B, B)
            if not folded_b:
                dAa, rows = dgrad(Lb, dzb, B, B, prev=La)
            if La.name in fold and La.cout == 64 and min(hk, wk) > 8:     # (bdn_conv3x3_dgrad_bb refuses maps of 8x8 and below)
                dza, dc = fold_dgrad(La, ptr(dAa), B, B, rows)
                wgrad(La, dza, ws.f[k], ck, ws.U[j], cprev, IN_PLAIN, None, B, B)
            else:
                dza = bn_bwd(La, ptr(dAa), La.cout, B, B, fused_rows=rows)
                # bf16x3: the operand is the split buffer the forward left (('a', layer)); the float32 skip / upsampled map do not exist
                wgrad(La, dza, None if self.x3 else ws.f[k], ck, None if self.x3 else ws.U[j], cprev, IN_PLAIN, None, B, B)
                dc = dgrad(La, dza, B, B)                   # [B,hk,wk, ck + cprev] = [dF_k | dU_j]
            dcat[k] = dc
            dprev = e(B, hs, wsrc, cprev)
            rows_up = _lib.load().bdn_upsample2x_bwd_rows(self.dt, B, hs, wsrc, cprev) if (j > 1 and FUSE_UPS_BS) else 0
            if rows_up:
                # the gradient lands on relu(bn(z)) of the previous decoder stage: its BatchNorm-backward partial sums come out of the same pass
                Lp = by[f'd{j - 1}b']
                call('bdn_upsample2x_bwd_bs', self.dt, dc.data_ptr() + ck * es, La.cin, ptr(dprev), ptr(ws.z[Lp.name]), ptr(ws.bn[Lp.name]),
                     ptr(ws.stats), B, hs, wsrc, hk, wk, cprev, st)
            else:
                call('bdn_upsample2x_bwd', self.dt, dc.data_ptr() + ck * es, La.cin, ptr(dprev), B, hs, wsrc, hk, wk, cprev, st)
            keep += [dzb, dAa, dza, dprev]
            if j > 1:
                dA_ptr, ldA = ptr(dprev), cprev
            else:
                dF5 = dprev
        # ---- encoder (both dates at once)
        dP = None
        for k in range(5, 0, -1):
            hk, wk = ws.dims[k - 1]
            La, Lb = by[f'e{k}a'], by[f'e{k}b']
            ck = ENC_CH[k - 1]
            if k == 5:
                dF_ptr, ldF = ptr(dF5), ck
            else:
                dF_ptr, ldF = ptr(dcat[k]), dcat[k].shape[3]
            rows_b = _lib.load().bdn_enc_skip_bwd_rows(self.dt, B, hk, wk, ck)
            dAb = e(2 * B, hk, wk, ck)
            call('bdn_enc_skip_bwd', self.dt, dF_ptr, ldF, ptr(ws.z[Lb.name]), ptr(ws.bn[Lb.name]),
                 ptr(dP), ptr(dAb), ptr(ws.stats), B, hk, wk, ck, st)
            if Lb.name in fold and min(hk, wk) > 8 and Lb.cout == 64:
                dzb, dAa, rows = fold_dgrad(Lb, ptr(dAb), 2 * B, B, rows_b, prev=La)
                wgrad(Lb, dzb, ws.z[La.name], Lb.cin, None, 0, IN_BNRELU, ws.bn[La.name], 2 * B, B)
            else:
                dzb = bn_bwd(Lb, ptr(dAb), ck, 2 * B, B, fused_rows=rows_b)
                wgrad(Lb, dzb, ws.z[La.name], Lb.cin, None, 0, IN_BNRELU, ws.bn[La.name], 2 * B, B)
                dAa, rows = dgrad(Lb, dzb, 2 * B, B, prev=La)
            if k == 1 and self.first_wgrad_fused and _lib.load().bdn_conv3x3_wgrad_bnbwd_supported(self.mdt, 2 * B, hk, wk, La.cout, La.cin, B):
                # the first conv has no data gradient: its dz has one reader, so the BatchNorm backward is applied inside
                # that weight-gradient GEMM's staging and the largest tensor of the step is never written (on the main
                # stream: nothing of the chain is left to run, the side stream is still busy with e1b's GEMM).
                # bf16x3 (round 6): float32 dA and z, the kernel splits dz into bf16 hi + lo in its staging and takes the input's split
                # operand the forward left; terms of the split product as engine.x3_bwd_terms
                call('bdn_bn_bwd_finalize', ptr(ws.bn[La.name]), 2, La.cout, ptr(ws.stats), rows, 1, ptr(sc['sums']),
                     ptr(grads[f'{La.bn}.weight']), ptr(grads[f'{La.bn}.bias']), ptr(ws.bnws), st)
                if self.x3:
                    fdt = BDN_BF16X2 if self.x3_bwd_terms == 2 else BDN_BF16X3
                    fin = ws.split_buf(('a', La.name), 2 * B * hk * wk * 2 * La.cin)
                else:
                    fdt, fin = self.dt, ws.x0
                wargs = (fdt, ptr(dAa), La.cout, ptr(ws.z[La.name]), ptr(ws.bn[La.name]), ptr(sc['sums']), B, La.cout,
                         ptr(fin), La.cin, ptr(sc['wg1']), ptr(grads[f'{La.conv}.weight']), La.cin_real, 2 * B, hk, wk, st)
                if not self._diag_skip_wgrad:
                    if self.prof is not None and (self.prof_filter is None or 'wgrad_first_kernel' in self.prof_filter):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        call('bdn_conv3x3_wgrad_bnbwd', *wargs)
                        e1.record()
                        self.prof.append(('wgrad_first_kernel', 2.0 * 2 * B * hk * wk * La.cout * 9 * La.cin, e0, e1))
                    else:
                        call('bdn_conv3x3_wgrad_bnbwd', *wargs)
                if zero_bias_grads:
                    grads[f'{La.conv}.bias'].zero_()
                if side is not None:
                    # this ready() may launch the LAST bucket's all-reduce, ordered behind the current (main) stream only;
                    # the bucket also holds e1b / e2a weight gradients whose GEMM + reduction are still queued on the side
                    # stream, so main joins side first (nothing of the chain is left to delay)
                    handoff(side, main)
                ready([f'{La.bn}.weight', f'{La.bn}.bias', f'{La.conv}.weight', f'{La.conv}.bias'])
                keep += [dAb, dzb, dAa, dP]
                dP = None
                continue
            src = ws.x0 if k == 1 else (None if self.x3 else ws.pool[k])
            if La.name in fold and k > 1 and min(hk, wk) > 8 and La.cout == 64:
                dza, dP_new = fold_dgrad(La, ptr(dAa), 2 * B, B, rows)
                wgrad(La, dza, src, La.cin, None, 0, IN_PLAIN, None, 2 * B, B)
                keep += [dAb, dzb, dAa, dza, dP]
                dP = dP_new
                continue
            dza = bn_bwd(La, ptr(dAa), La.cout, 2 * B, B, fused_rows=rows)
            wgrad(La, dza, src, La.cin, None, 0, IN_PLAIN, None, 2 * B, B, on_main=(k == 1 and self.last_wgrad_on_chain))
            keep += [dAb, dzb, dAa, dza, dP]
            dP = dgrad(La, dza, 2 * B, B) if k > 1 else None
        if side is not None:
            handoff(side, main)                      # every weight gradient is complete before the caller's next kernel
        return grads
