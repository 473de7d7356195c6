"""Host-side engine: owns a dod_handle, pushes the module's parameters across the C ABI,
keeps a workspace tensor, and launches the forward on torch's current HIP stream.

The engine is the only place that touches libdinodet.so; torch provides device memory and the
stream, nothing else.  No CPU path exists: a non-GPU tensor raises.
"""
import ctypes as C
import os

import torch

from . import _native as nat
from .config import BackboneConfig, DecoderConfig


def default_precision():
    """Default = the parity-gated mode on the bf16 matrix cores ("bf16x3": within 1e-3 of the reference's fp32 CPU
    forward, tests/test_gpu_forward.py GATED).  "bf16" (single-pass bf16 operands: fastest, ~2e-2 from the reference on
    logits) and "fp8" are opt-in throughput modes; "fp32" is the exact-fp32 mode."""
    return os.environ.get("DINODET_PRECISION", "bf16x3")


def make_config(bb: BackboneConfig, dc: DecoderConfig, precision: str) -> nat.DodConfig:
    if precision not in nat.PREC:
        raise ValueError(f"precision must be one of {sorted(nat.PREC)}, got {precision!r}")
    c = nat.DodConfig()
    c.hidden, c.layers, c.heads, c.swiglu = bb.hidden, bb.layers, bb.heads, int(bb.swiglu)
    c.patch, c.pos_grid, c.ffn_hidden, c.ln_eps = bb.patch, bb.pos_grid, bb.ffn_hidden, bb.ln_eps
    c.lora_r, c.lora_alpha, c.target_dim = bb.lora_r, bb.lora_alpha, bb.target_dim
    c.num_queries, c.dec_hidden, c.dec_heads, c.dec_layers = dc.num_queries, dc.hidden_dim, dc.nheads, dc.num_layers
    c.num_classes, c.dim_feedforward, c.n_points = dc.num_classes, dc.dim_feedforward, dc.n_points
    c.use_deformable, c.dec_ln_eps = int(dc.use_deformable), dc.ln_eps
    c.precision = nat.PREC[precision]
    return c


class Engine:
    def __init__(self, bb: BackboneConfig, dc: DecoderConfig, precision: str):
        self.bb, self.dc, self.precision = bb, dc, precision
        self._lib = nat.lib()
        self._h = C.c_void_p()
        cfg = make_config(bb, dc, precision)
        nat.check(self._lib.dod_create(C.byref(cfg), C.byref(self._h)))
        self._sig = None
        self._ws = None
        self._device = None  # the device the packed weights live on (set by sync_weights)
        self._keep = []     # tensors whose pointers the handle holds until finalize returns
        self._tap_bufs = {}
        # hipGraph replay of the whole forward, one captured graph per input shape (opt-in: DINODET_HIPGRAPH=1 or
        # model.enable_hipgraph()): ~100 launches per forward are launch-bound at small batch
        self.use_graph = os.environ.get("DINODET_HIPGRAPH", "0") == "1"
        self._graphs = {}   # (shape, device) -> (graph, static input, static output, the workspace(s) the graph was captured on)
        # Graph replays of a large batch run it as `micro_streams` concurrent micro-batches on separate HIP streams (fork / join
        # inside the captured graph, one workspace each, one shared output buffer): images are independent, and one micro-batch's
        # LayerNorm / epilogue / last-round bubbles are filled by the other's GEMMs -- measured +4-5 % at batch 64 x 518^2 with
        # bit-identical detections (tools/bench_two_streams.py); four micro-batches lose (1 993 vs 2 167 images/s), three too (round 4: 2 318
        # vs 2 433), a start skew between the streams (13 us .. 1.2 ms) changes nothing, and neither do uneven halves (34 / 30 images: 2 435 vs
        # 2 428; 36 / 28: 2 385; 40 / 24: 2 368 -- tools/experiments/r4_exp31.sh).  DINODET_MICRO_STREAMS=1 switches it off.
        self.micro_streams = max(1, int(os.environ.get("DINODET_MICRO_STREAMS", "2")))
        # threshold in TOKEN ROWS of the batch (images x tokens), not images: 16 images of 518^2 (ViT-L: 21 920 rows) gain 7 %, 8
        # images of 518^2 (10 960 rows: BASELINE configs[2] sharded over 8 GPUs) 10 %, 32 images of 224^2 (8 224 rows) nothing, 32
        # images of 518^2 on ViT-g nothing (measured)
        self.micro_min_rows = int(os.environ.get("DINODET_MICRO_MIN_ROWS", "10000"))
        self.micro_min_batch = int(os.environ.get("DINODET_MICRO_MIN_BATCH", "2"))
        self._micro = {}    # (micro-batch, H, W, device) -> (workspaces, side streams) of the eager / caller-captured path

    def close(self):
        if self._h:
            self._lib.dod_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ weights
    @staticmethod
    def signature(named):
        return tuple((k, t.data_ptr(), t._version, tuple(t.shape)) for k, t in named)

    def sync_weights(self, named):
        """named: list of (reference state-dict key, fp32 CUDA tensor).  Re-packs only when a
        tensor moved or was modified in place (load_state_dict, optimizer step, .to())."""
        sig = self.signature(named)
        if sig == self._sig:
            return
        self._keep = []
        devs = {t.device for _, t in named}
        if len(devs) != 1:
            raise RuntimeError(f"parameters live on several devices ({sorted(map(str, devs))}): one replica per GPU")
        dev = next(iter(devs))
        if dev.type != "cuda":
            raise RuntimeError(f"parameters are on {dev}: the MI355X path has no CPU fallback")
        with torch.cuda.device(dev):        # hipMalloc of the packed copy and the pack kernels bind to the parameters' GPU
            for k, t in named:
                if t.dtype == torch.bfloat16 and t.is_contiguous():
                    dt = nat.DOD_BF16
                else:
                    dt = nat.DOD_F32
                    if t.dtype != torch.float32 or not t.is_contiguous():
                        t = t.detach().to(torch.float32).contiguous()
                        self._keep.append(t)
                shape = (C.c_int64 * max(t.dim(), 1))(*t.shape)
                nat.check(self._lib.dod_set_weight(self._h, k.encode(), nat.ptr(t), shape, t.dim(), dt), self._h)
            nat.check(self._lib.dod_finalize_weights(self._h, nat.stream_ptr()), self._h)
        self._keep = []
        self._sig = sig
        self._device = dev
        self._ws = None
        self._micro = {}
        self._graphs = {}       # packed weights were reallocated: captured graphs hold stale pointers

    # ------------------------------------------------------------------ forward
    def _workspace(self, nbytes, device):
        """the eager calls' scratch; captured graphs own theirs (a graph bakes the addresses in, so its workspace must
        outlive every later, larger eager call)"""
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != device:
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        return self._ws

    def _on_device(self, x):
        """input and packed weights must share a GPU; the returned context makes that GPU current (streams, hipMalloc)"""
        if self._device is not None and x.device != self._device:
            raise RuntimeError(f"input is on {x.device} but the model's parameters are on {self._device}")
        return torch.cuda.device(x.device)

    @staticmethod
    def _check_pixels(x):
        if x.dim() != 4:
            raise ValueError(f"pixel_values must be [batch, 3, H, W], got shape {tuple(x.shape)}")
        if x.shape[1] != 3:
            # same message as Dinov2PatchEmbeddings.forward (modeling_dinov2.py:143-147)
            raise ValueError("Make sure that the channel dimension of the pixel values match with the one set in "
                             f"the configuration. Expected 3 but got {x.shape[1]}.")
        if not x.is_cuda:
            raise RuntimeError("pixel_values must be on the GPU: the MI355X path has no CPU fallback")
        return x.detach().to(torch.float32).contiguous()

    def _forward_graph(self, x):
        """replay (capturing on first use) the forward for x's shape; returns the graph's static output buffer"""
        key = (tuple(x.shape), x.device)
        ent = self._graphs.get(key)
        if ent is None:
            xs = x.clone()
            B, _, H, W = x.shape
            K = self.micro_streams if (self._micro_ok(B, H, W) and B % self.micro_streams == 0) else 1
            Bk = B // K
            parts = [(xs[k * Bk:(k + 1) * Bk], torch.empty(self._lib.dod_workspace_bytes(self._h, Bk, H, W), dtype=torch.uint8, device=x.device))
                     for k in range(K)]
            out = torch.empty(B, self.dc.num_queries, self.dc.num_classes + 4, dtype=torch.float32, device=x.device)
            for k, (xk, wk) in enumerate(parts):           # warm-up outside capture: position table, func attributes
                self._launch_forward(xk, wk, out[k * Bk:(k + 1) * Bk])
            torch.cuda.synchronize(x.device)
            side = [torch.cuda.Stream(device=x.device) for _ in range(K - 1)]
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                cur = torch.cuda.current_stream(x.device)
                for st in side:
                    st.wait_stream(cur)                     # fork
                self._launch_forward(parts[0][0], parts[0][1], out[:Bk])
                for k, st in enumerate(side, start=1):
                    with torch.cuda.stream(st):
                        self._launch_forward(parts[k][0], parts[k][1], out[k * Bk:(k + 1) * Bk])
                for st in side:
                    cur.wait_stream(st)                     # join
            ent = (g, xs, out, [w for _, w in parts], side)
            self._graphs[key] = ent
        g, xs, out = ent[0], ent[1], ent[2]
        xs.copy_(x)
        g.replay()
        return out

    def _launch_forward(self, x, ws=None, det=None):
        B, _, H, W = x.shape
        nbytes = self._lib.dod_workspace_bytes(self._h, B, H, W)
        if nbytes == 0:
            raise ValueError(f"unsupported input {tuple(x.shape)}")
        if ws is None:
            ws = self._workspace(nbytes, x.device)
        if det is None:
            det = torch.empty(B, self.dc.num_queries, self.dc.num_classes + 4, dtype=torch.float32, device=x.device)
        nat.check(self._lib.dod_forward(self._h, nat.ptr(x), B, H, W, nat.ptr(det), nat.ptr(ws), ws.numel(),
                                        nat.stream_ptr()), self._h)
        return det

    def _micro_ok(self, B, H, W):
        K = self.micro_streams
        return (K >= 2 and B >= max(self.micro_min_batch, K)
                and B * self._lib.dod_num_tokens(self._h, H, W) >= self.micro_min_rows)

    def _launch_micro(self, x):
        """the forward of a large batch as `micro_streams` concurrent micro-batches (fork / join on side streams; inside a caller's
        stream capture the side streams join the capture).  Detections are bit-identical to the single launch."""
        B, _, H, W = x.shape
        K = self.micro_streams
        if not self._micro_ok(B, H, W) or self._tap_bufs:
            return self._launch_forward(x)
        sizes = [B // K + (1 if k < B % K else 0) for k in range(K)]
        offs = [sum(sizes[:k]) for k in range(K)]
        key = (tuple(sizes), H, W, x.device)
        st = self._micro.get(key)
        if st is None:
            wss = []
            for bk in sizes:
                nb = self._lib.dod_workspace_bytes(self._h, bk, H, W)
                if nb == 0:
                    raise ValueError(f"unsupported input {tuple(x.shape)}")
                wss.append(torch.empty(nb, dtype=torch.uint8, device=x.device))
            st = (wss, [torch.cuda.Stream(device=x.device) for _ in range(K - 1)])
            self._micro[key] = st
        wss, side = st
        det = torch.empty(B, self.dc.num_queries, self.dc.num_classes + 4, dtype=torch.float32, device=x.device)
        cur = torch.cuda.current_stream(x.device)
        # the (H, W) position table is built (hipMalloc + resize kernel) by the first forward at a new grid, on the stream of THAT
        # launch: done here on `cur`, before the fork, so that the side streams' patch embeds are ordered behind the resize kernel
        # (a cache hit is a host-side lookup)
        nat.check(self._lib.dod_prepare(self._h, H, W, nat.stream_ptr()), self._h)
        for sd in side:
            sd.wait_stream(cur)                              # fork: x (and det's allocation) are ready on `cur`
        self._launch_forward(x[:sizes[0]], wss[0], det[:sizes[0]])
        for k, sd in enumerate(side, start=1):
            with torch.cuda.stream(sd):
                self._launch_forward(x[offs[k]:offs[k] + sizes[k]], wss[k], det[offs[k]:offs[k] + sizes[k]])
        for sd in side:
            cur.wait_stream(sd)                              # join
        return det

    def forward(self, pixel_values, named):
        x = self._check_pixels(pixel_values)
        self.sync_weights(named)
        with self._on_device(x):
            if self.use_graph and not torch.cuda.is_current_stream_capturing() and not self._tap_bufs:
                return self._forward_graph(x)
            return self._launch_micro(x)

    def forward_u8(self, pixels_hwc, named):
        """uint8 [B, H, W, 3] (preprocess_batch(..., as_uint8=True)) -> packed detections; ToTensor's / 255 happens in the patch
        embedding's load stage (dod_forward_u8)"""
        x = pixels_hwc
        if x.dim() != 4 or x.shape[-1] != 3 or x.dtype != torch.uint8:
            raise ValueError(f"expected uint8 [batch, H, W, 3], got {x.dtype} {tuple(x.shape)}")
        if not x.is_cuda:
            raise RuntimeError("pixels must be on the GPU: the MI355X path has no CPU fallback")
        x = x.contiguous()
        self.sync_weights(named)
        B, H, W, _ = x.shape
        nbytes = self._lib.dod_workspace_bytes(self._h, B, H, W)
        if nbytes == 0:
            raise ValueError(f"unsupported input {tuple(x.shape)}")
        with self._on_device(x):
            ws = self._workspace(nbytes, x.device)
            det = torch.empty(B, self.dc.num_queries, self.dc.num_classes + 4, dtype=torch.float32, device=x.device)
            nat.check(self._lib.dod_forward_u8(self._h, nat.ptr(x), B, H, W, nat.ptr(det), nat.ptr(ws), ws.numel(), nat.stream_ptr()), self._h)
        return det

    def backbone_forward(self, pixel_values, named):
        x = self._check_pixels(pixel_values)
        self.sync_weights(named)
        B, _, H, W = x.shape
        nbytes = self._lib.dod_workspace_bytes(self._h, B, H, W)
        if nbytes == 0:
            raise ValueError(f"unsupported input {tuple(x.shape)}")
        with self._on_device(x):
            ws = self._workspace(nbytes, x.device)
            N = self._lib.dod_num_tokens(self._h, H, W)
            feats = torch.empty(B, N, self.bb.out_dim, dtype=torch.float32, device=x.device)
            nat.check(self._lib.dod_backbone_forward(self._h, nat.ptr(x), B, H, W, nat.ptr(feats), nat.ptr(ws),
                                                     ws.numel(), nat.stream_ptr()), self._h)
        return feats

    def backbone_prefix(self, pixel_values, named, nblocks):
        """embeddings + the first `nblocks` encoder blocks -> fp32 residual stream [B, N, hidden] (no autograd)"""
        x = self._check_pixels(pixel_values)
        self.sync_weights(named)
        B, _, H, W = x.shape
        nbytes = self._lib.dod_workspace_bytes(self._h, B, H, W)
        if nbytes == 0:
            raise ValueError(f"unsupported input {tuple(x.shape)}")
        with self._on_device(x):
            ws = self._workspace(nbytes, x.device)
            N = self._lib.dod_num_tokens(self._h, H, W)
            out = torch.empty(B, N, self.bb.hidden, dtype=torch.float32, device=x.device)
            nat.check(self._lib.dod_backbone_prefix(self._h, nat.ptr(x), B, H, W, int(nblocks), nat.ptr(out), nat.ptr(ws), ws.numel(),
                                                    nat.stream_ptr()), self._h)
        return out

    def decoder_forward(self, memory, named):
        if memory.dim() != 3 or memory.shape[-1] != self.dc.hidden_dim:
            raise ValueError(f"src must be [batch, seq_len, {self.dc.hidden_dim}], got {tuple(memory.shape)}")
        if not memory.is_cuda:
            raise RuntimeError("src must be on the GPU: the MI355X path has no CPU fallback")
        m = memory.detach().to(torch.float32).contiguous()
        self.sync_weights(named)
        B, N, _ = m.shape
        nbytes = self._lib.dod_decoder_workspace_bytes(self._h, B, N)
        with self._on_device(m):
            ws = self._workspace(nbytes, m.device)
            det = torch.empty(B, self.dc.num_queries, self.dc.num_classes + 4, dtype=torch.float32, device=m.device)
            nat.check(self._lib.dod_decoder_forward(self._h, nat.ptr(m), B, N, nat.ptr(det), nat.ptr(ws), ws.numel(),
                                                    nat.stream_ptr()), self._h)
        return det

    # ------------------------------------------------------------------ per-kernel timing (bench roofline leg)
    PROFILE_CLASSES = {"gemm_bf16": 0, "attn_bf16": 1, "gemm_f32": 2, "attn_f32": 3, "layernorm": 4, "gemm_fp8": 6}

    def profile(self, enable):
        nat.check(self._lib.dod_profile(self._h, int(bool(enable))), self._h)

    def profile_read(self):
        out = {}
        for name, cls in self.PROFILE_CLASSES.items():
            ms, fl, n = C.c_double(), C.c_double(), C.c_int()
            nat.check(self._lib.dod_profile_read(self._h, cls, C.byref(ms), C.byref(fl), C.byref(n)), self._h)
            out[name] = {"ms": ms.value, "flops": fl.value, "launches": n.value}
        return out

    # ------------------------------------------------------------------ debug taps (parity tests)
    def set_tap(self, stage, shape, device):
        buf = torch.zeros(*shape, dtype=torch.float32, device=device)
        self._tap_bufs[stage] = buf
        nat.check(self._lib.dod_set_tap(self._h, stage, nat.ptr(buf)), self._h)
        return buf

    def clear_taps(self):
        for st in list(self._tap_bufs):
            nat.check(self._lib.dod_set_tap(self._h, st, C.c_void_p(0)), self._h)
        self._tap_bufs = {}


def split_detections(det, num_classes):
    """packed [B,Q,C+4] -> the reference's output dict (detr_decoder.py:83): zero-copy views."""
    return {"pred_logits": det[..., :num_classes], "pred_boxes": det[..., num_classes:]}
