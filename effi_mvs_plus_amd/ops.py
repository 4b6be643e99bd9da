"""Tensor-level wrappers over the C ABI.  PyTorch is used for device memory and the current stream only.

All functions take fp32 CUDA (ROCm) tensors without a batch dimension and enqueue on the current
stream.  CPU tensors raise: there is no fallback path.
"""
from __future__ import annotations

import ctypes as C
import functools
import os
import threading

import weakref

import torch

from . import _lib
from ._lib import EffiLibraryError, check

ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH = 0, 1, 2, 3
EPI_PLAIN, EPI_GRU_ZR, EPI_GRU_Q, EPI_HEAD, EPI_ADD_UP2, EPI_NHWC = 0, 1, 2, 3, 4, 5
EPI_ADD_SHUF2, EPI_NHWC_ADD_SHUF2 = 9, 10      # split-precision 3x3 entry only: + pixel-shuffled coarser map (aux0 [4*cout,h/2,w/2])
MAX_VIEWS = 12


class KernelProfile:
    """Optional per-launch timing (HIP events on the launching stream) used by bench.py for the roofline
    line: ``keys`` selects which launches are bracketed (None = all); ``records`` collects
    (key, algorithmic work, start event, end event)."""

    def __init__(self, keys=None):
        self.keys = None if keys is None else set(keys)
        self.records = []

    def busy_ms_per_mark(self):
        """Sum of the launches' own durations between consecutive ``ops.mark`` calls (name of the mark that ENDS the span -> ms):
        what the span costs when its kernels run back to back, as in a graph replay -- independent of how fast the host enqueues."""
        torch.cuda.synchronize()
        out, acc = {}, 0.0
        for key, _, e0, e1 in self.records:
            if e0 is None:
                out[key] = out.get(key, 0.0) + acc
                acc = 0.0
            else:
                acc += e0.elapsed_time(e1)
        return out

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for key, work, e0, e1 in self.records:
            if e0 is None:                      # a stage mark (see ``mark``), not a launch
                continue
            d = out.setdefault(key, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            d["launches"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += work.get("flops", 0.0)
            d["bytes"] += work.get("bytes", 0.0)
        return out


_PROF = None

# Arithmetic of the 3x3 MFMA convolutions: "split" = every fp32 product as hi*hi + hi*lo + lo*hi on the bf16 MFMA with
# fp32 accumulation (~5e-6 of the output scale), "fp32" = v_mfma_f32_16x16x4_f32 (exact products).  Everything else on the
# path (warps, 3-D regularisation, soft-argmin, lookups, epilogues) is plain fp32 in both modes.
# "bf16" = the same kernels as "split" compiled with ONLY the hi*hi term (entry points *_bf16): plain bf16 operands, fp32 accumulation --
# BASELINE.json's "bf16 (MFMA 3D-conv path)" configuration.  It is NOT fp32-grade: its own tolerance is a normalised mean depth
# error <= 1e-2; never the default, never the headline.
PRECISIONS = ("split", "fp32", "bf16")
_PRECISION = os.environ.get("EFFI_MVS_PRECISION", "split")
if _PRECISION not in PRECISIONS:
    raise ValueError(f"EFFI_MVS_PRECISION must be one of {PRECISIONS}, got {_PRECISION!r}")


def set_precision(mode):
    global _PRECISION
    if mode not in PRECISIONS:
        raise ValueError(f"precision must be one of {PRECISIONS}, got {mode!r}")
    _PRECISION = mode


def get_precision():
    return _PRECISION


def uses_split():
    """True when the 3x3 / 3-D MFMA convolutions run on the bf16 matrix cores (split-precision products or bf16 operands)."""
    return _PRECISION in ("split", "bf16")


def _x3(name):
    """The split-precision entry ``name`` of the library, or its plain-bf16-operand twin (suffix _bf16) in "bf16" precision."""
    return getattr(_lib.lib(), name + "_bf16" if _PRECISION == "bf16" else name)


# A/B switches.  Python-level ones (which fused form a module launches) live in ``_PY_OPTS``; kernel-level ones (tile rules, kernel
# forms) in the library's own table (include/effi_mvs_hip.h: effi_set_option).  Both are initialised ONCE from the environment
# (EFFI_<NAME>) and changed afterwards through ``set_option`` -- nothing on a per-call path reads the environment.
_PY_OPTION_DEFAULTS = {"warp_x3": 0, "state_q4": 1, "c1k7_mfma": 1, "k5s2_split": 1, "roll": 1, "conv3d_unaligned_split": 1, "conv3d_s2_split": 1, "fpn_conv0_fused": 1,
                       "fpn_split_head": 1, "csp_pair": 1, "head_taps": 1, "enc_tail": 0, "enc_gen": 1, "reduce_chunk": 2048, "gru_fused": 0, "csp_gen": 0}
_PY_OPTS = {k: int(os.environ.get("EFFI_" + k.upper(), v)) for k, v in _PY_OPTION_DEFAULTS.items()}
LIB_OPTIONS = ("warp_lds_kb", "dyn_form", "dyn_setup_exact", "dyn_xchg", "pixnet_mfma", "force_mr", "mr4_min", "mr4_nt2_max", "mr2_min",
               "wide_tiles", "roll_mr", "roll_zt", "roll_rp", "deconv_mr", "sr_waves", "enc_gen_mr3", "c3_lean", "dyn_win")


def option(name):
    """Current value of a Python-level switch."""
    return _PY_OPTS[name]


def get_option(name):
    """Value of any switch (None = unset library switch: the built-in rule applies)."""
    if name in _PY_OPTS:
        return _PY_OPTS[name]
    if name not in LIB_OPTIONS:
        raise KeyError(name)
    L = _lib.lib()
    v = L.effi_get_option(name.encode())
    return None if v == L.effi_option_unset() else int(v)


def set_option(name, value):
    """Set a switch (``value=None``: back to the default / unset).  Returns the previous value (for restoring)."""
    before = get_option(name)
    if name in _PY_OPTS:
        _PY_OPTS[name] = _PY_OPTION_DEFAULTS[name] if value is None else int(value)
    else:
        L = _lib.lib()
        check(L.effi_set_option(name.encode(), L.effi_option_unset() if value is None else int(value)), "effi_set_option")
    return before


class options:
    """``with ops.options(warp_lds_kb=0, head_taps=0): ...`` -- switches set for the block, restored after it."""

    def __init__(self, **kw):
        self.kw, self.before = kw, {}

    def __enter__(self):
        for k, v in self.kw.items():
            self.before[k] = set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.before.items():
            set_option(k, v)
        return False


def set_profile(p):
    global _PROF
    _PROF = p


def get_profile():
    return _PROF


def _call(key, work, fn, *args):
    p = _PROF
    if p is None or (p.keys is not None and key not in p.keys):
        return fn(*args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = fn(*args)
    e1.record()
    p.records.append((key, work(), e0, e1))
    return rc


# Caller-owned per-device workspace of the C ABI (include/effi_mvs_hip.h: effi_workspace_bytes / effi_set_workspace): a small
# zero-filled block per device ordinal, allocated here through torch the first time a tensor of that device reaches an op and
# kept alive for the life of the process.  The library itself allocates nothing.
_WORKSPACES = {}
_WS_LOCK = threading.Lock()


def ensure_workspace(index):
    """Register the zero-filled workspace of device ``index`` with the library (idempotent)."""
    if index in _WORKSPACES:
        return
    with _WS_LOCK:
        if index in _WORKSPACES:
            return
        if torch.cuda.is_current_stream_capturing():
            raise EffiLibraryError(f"cuda:{index}: the device's workspace must be registered before stream capture starts "
                                   "(run one pass eagerly, or call ops.ensure_workspace(index), before capturing)")
        L = _lib.lib()
        n = int(L.effi_workspace_bytes())
        with torch.cuda.device(index):
            ws = torch.zeros((n + 3) // 4, device=torch.device("cuda", index), dtype=torch.float32)
            torch.cuda.current_stream().synchronize()       # zero-filled before ANY stream of the device can read it
        check(L.effi_set_workspace(index, C.c_void_p(ws.data_ptr()), n), "effi_set_workspace")
        _WORKSPACES[index] = ws


def _t(x: torch.Tensor, name: str, contiguous=True) -> torch.Tensor:
    if not isinstance(x, torch.Tensor):
        raise TypeError(f"{name}: expected a tensor")
    if not x.is_cuda:
        raise EffiLibraryError(f"{name}: CPU tensor passed to the HIP path (no CPU fallback exists)")
    idx = x.device.index
    if idx != torch._C._cuda_getDevice():
        # kernels are enqueued on the CURRENT device's stream: a tensor of another device would be a wild pointer there
        raise EffiLibraryError(f"{name}: tensor lives on cuda:{idx} but the current device is cuda:{torch._C._cuda_getDevice()}; "
                               "call inside `with torch.cuda.device(tensor.device):` (the public modules do this themselves)")
    if idx not in _WORKSPACES:
        ensure_workspace(idx)
    if x.dtype != torch.float32:
        raise TypeError(f"{name}: fp32 only (got {x.dtype}); the reference path is fp32 (models/module.py:318)")
    if contiguous and not x.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    return x


def _p(x):
    return C.c_void_p(x.data_ptr()) if x is not None else C.c_void_p(0)


def _first_cuda_tensor(obj, depth=0):
    if isinstance(obj, torch.Tensor):
        return obj if obj.is_cuda else None
    if depth < 3:
        if isinstance(obj, dict):
            obj = obj.values()
        if isinstance(obj, (list, tuple)) or type(obj).__name__ == "dict_values":
            for o in obj:
                t_ = _first_cuda_tensor(o, depth + 1)
                if t_ is not None:
                    return t_
    return None


def on_tensor_device(fn):
    """Decorator of the public entry points (module ``forward``s, the reference-named functions): run with the device of the
    first CUDA tensor among the arguments as the current device, as stock PyTorch operators do.  The kernels are enqueued on the
    current device's stream, so a model on cuda:1 called while cuda:0 is current would otherwise launch against foreign
    pointers.  (Replicas of nn.DataParallel already run under their own device.)"""
    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        t_ = _first_cuda_tensor(args)
        if t_ is None and kwargs:
            t_ = _first_cuda_tensor(kwargs)
        if t_ is None or t_.device.index == torch._C._cuda_getDevice():
            return fn(*args, **kwargs)
        with torch.cuda.device(t_.device):
            return fn(*args, **kwargs)
    return wrapper


# Independent kernel chains (the mask head, the preparation of the stages' GRU inputs, the pyramid passes of the odd views) CAN
# be enqueued on a second HIP stream (``Branch``).  Off by default: inside a captured graph every cross-stream edge costs ~5 us
# (fork) / ~11 us (join) of idle GPU, and a graph with ANY such edge replays ~3 % slower than the linear graph of the same
# kernels (measured at 1600x1184: 2.64 -> 2.57 ms; at 800x576: 1.40 -> 1.31 ms) -- the chains that do overlap well share ONE
# grid instead (effi_encoder_inputs_f32, effi_*_pair_f32).  A second stream pays off when several views are in flight at once
# (bench.py --in-flight 3: 2.31 ms per view with branches, 2.43 without).  EFFI_MVS_BRANCHES=1 / set_branches(True) turn it on;
# results are identical either way (no atomics; every kernel sees its producers through stream events).
_BRANCHES = os.environ.get("EFFI_MVS_BRANCHES", "0") != "0"
_SIDE_STREAMS = {}


def set_branches(enabled):
    """Enable / disable the second stream (profiling passes want serial kernel durations)."""
    global _BRANCHES
    _BRANCHES = bool(enabled)


def get_branches():
    return _BRANCHES


class Branch:
    """``with Branch() as br: <enqueue the side chain>`` ... main chain ... ``br.join(outputs...)``.

    The side stream first waits for everything already enqueued on the current (main) stream; ``join`` makes the main
    stream wait for the side chain and tells the caching allocator that the given side-allocated tensors are now used on
    the main stream."""

    def __init__(self):
        self.enabled = _BRANCHES and torch.cuda.is_available()      # CPU tensors must reach the ops' own loud failure
        if self.enabled:
            self.main = torch.cuda.current_stream()
            key = (self.main.device.index, self.main.cuda_stream)
            side = _SIDE_STREAMS.get(key)
            if side is None:
                side = _SIDE_STREAMS[key] = torch.cuda.Stream(device=self.main.device)
            self.side = side
            self._ctx = None

    def __enter__(self):
        if self.enabled:
            self.side.wait_stream(self.main)
            self._ctx = torch.cuda.stream(self.side)
            self._ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.enabled:
            self._ctx.__exit__(*exc)
        return False

    def join(self, *tensors):
        if self.enabled:
            self.main.wait_stream(self.side)
            for t_ in tensors:
                if t_ is not None:
                    t_.record_stream(self.main)


# Optional stage markers (bench.py's "ms per cost-volume stage"): forward_hot calls mark() at the stage boundaries; events are only
# recorded while a list is installed (never during graph capture).
_MARKS = None


def set_marks(marks):
    global _MARKS
    _MARKS = marks


def mark(name):
    if _PROF is not None and _PROF.keys is None:
        _PROF.records.append((name, {}, None, None))
    if _MARKS is not None:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        _MARKS.append((name, ev))


def _stream():
    # raw hipStream_t of torch's current stream on the current device (fast path: ~1 us instead of the
    # ~8 us of torch.cuda.current_stream().cuda_stream; this is called once per kernel launch)
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


def _ptr_array(tensors):
    return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def _int_array(vals):
    return (C.c_int * len(vals))(*vals)


# ---------------------------------------------------------------------------------------------
def compose_rel_proj(pairs: torch.Tensor) -> torch.Tensor:
    """pairs [N,2,4,4] -> rt [N-1,12]  (K.[R|t] then P_src . P_ref^-1)."""
    _t(pairs, "pairs")
    n = pairs.shape[0]
    rt = torch.empty(n - 1, 12, device=pairs.device, dtype=torch.float32)
    check(_lib.lib().effi_compose_rel_proj_f32(_p(pairs), n, _p(rt), _stream()), "effi_compose_rel_proj_f32")
    return rt


def compose_rel_proj_stages(pairs_list):
    """``compose_rel_proj`` of up to 4 stages' [N,2,4,4] tensors in one launch -> list of rt [N-1,12] (views of one buffer)."""
    for p_ in pairs_list:
        _t(p_, "pairs")
    n = pairs_list[0].shape[0]
    if len(pairs_list) > 4 or any(p_.shape[0] != n for p_ in pairs_list):
        raise ValueError("compose_rel_proj_stages: up to 4 stages with the same number of views")
    rt = torch.empty(len(pairs_list), n - 1, 12, device=pairs_list[0].device, dtype=torch.float32)
    check(_lib.lib().effi_compose_rel_proj_stages_f32(_ptr_array(pairs_list), len(pairs_list), n, _p(rt), _stream()),
          "effi_compose_rel_proj_stages_f32")
    return [rt[k] for k in range(len(pairs_list))]


def rel_proj(src_proj: torch.Tensor, ref_proj: torch.Tensor) -> torch.Tensor:
    _t(src_proj, "src_proj"), _t(ref_proj, "ref_proj")
    rt = torch.empty(12, device=src_proj.device, dtype=torch.float32)
    check(_lib.lib().effi_rel_proj_f32(_p(src_proj), _p(ref_proj), _p(rt), _stream()), "effi_rel_proj_f32")
    return rt


def to_nhwc(feats):
    """list of planar [C,h,w] maps -> list of channel-last [h,w,C] maps.  A map that is already
    channel-last in memory (torch.channels_last) is passed through without a copy."""
    out, todo_src, todo_dst = [None] * len(feats), [], []
    for i, f in enumerate(feats):
        _t(f, "feature", contiguous=False)
        C_, h, w = f.shape
        if f.permute(1, 2, 0).is_contiguous() and C_ > 1:
            out[i] = f.permute(1, 2, 0)
        else:
            if not f.is_contiguous():
                raise ValueError("feature map must be planar-contiguous or channels-last")
            d = torch.empty(h, w, C_, device=f.device, dtype=torch.float32)
            out[i] = d
            todo_src.append(f)
            todo_dst.append(d)
    if todo_src:
        C_, h, w = todo_src[0].shape
        for k in range(0, len(todo_src), MAX_VIEWS + 1):
            s, d = todo_src[k:k + MAX_VIEWS + 1], todo_dst[k:k + MAX_VIEWS + 1]
            check(_lib.lib().effi_planar_to_nhwc_f32(_ptr_array(s), _ptr_array(d), len(s), C_, h * w, _stream()),
                  "effi_planar_to_nhwc_f32")
    return out


def _depth_strides(depth: torch.Tensor, D, h, w):
    """depth hypotheses given as [D] (uniform), [D,h,w] contiguous, or an expanded [D,h,w] view."""
    if depth.dim() == 1:
        return depth, 1, 0
    if depth.stride(1) == 0 and depth.stride(2) == 0:
        return depth, depth.stride(0), 0
    if not depth.is_contiguous():
        depth = depth.contiguous()
    return depth, h * w, 1


def homo_warp(src_nhwc, rt, depth, D):
    h, w, Cc = src_nhwc.shape
    _t(src_nhwc, "src_nhwc"), _t(rt, "rt"), _t(depth, "depth", contiguous=False)
    depth, dds, dps = _depth_strides(depth, D, h, w)
    out = torch.empty(Cc, D, h, w, device=src_nhwc.device, dtype=torch.float32)
    check(_lib.lib().effi_homo_warp_f32(_p(src_nhwc), _p(rt), _p(depth), dds, dps, Cc, h, w, D, _p(out), _stream()),
          "effi_homo_warp_f32")
    return out


def homo_warp_bwd(rt, depth, D, grad_out, h, w):
    """Backward of ``homo_warp`` w.r.t. the source features: grad_out [C,D,h,w] -> grad_src [h,w,C] (scope row n2)."""
    _t(rt, "rt"), _t(depth, "depth", contiguous=False), _t(grad_out, "grad_out")
    Cc = grad_out.shape[0]
    depth, dds, dps = _depth_strides(depth, D, h, w)
    g_src = torch.zeros(h, w, Cc, device=grad_out.device, dtype=torch.float32)
    check(_lib.lib().effi_homo_warp_bwd_f32(_p(rt), _p(depth), dds, dps, Cc, h, w, D, _p(grad_out), _p(g_src), _stream()),
          "effi_homo_warp_bwd_f32")
    return g_src


def warpcorr_views(ref_nhwc, srcs_nhwc, rt, depth, D, x3=False):
    """-> (sim_views [S,D,h,w], entropy [S,h,w]).  ``x3``: in "split" / "bf16" precision use the matrix-core form
    (``effi_warpcorr_views_x3_f32``: correlations of the tap pixels as split-precision MFMAs, then interpolated) -- inference only;
    the default is the exact fp32 kernel (what training and the exact-fp32 precision use)."""
    h, w, Cc = ref_nhwc.shape
    S = len(srcs_nhwc)
    _t(ref_nhwc, "ref_nhwc"), _t(rt, "rt"), _t(depth, "depth", contiguous=False)
    for s in srcs_nhwc:
        _t(s, "src_nhwc")
        if tuple(s.shape) != (h, w, Cc):
            raise ValueError("source / reference feature shapes differ")
    if rt.shape[0] != S:
        raise ValueError("Different number of images and projection matrices")
    depth, dds, dps = _depth_strides(depth, D, h, w)
    sim = torch.empty(S, D, h, w, device=ref_nhwc.device, dtype=torch.float32)
    ent = torch.empty(S, h, w, device=ref_nhwc.device, dtype=torch.float32)
    work = lambda: {"flops": S * D * h * w * (10.0 * Cc + 20), "bytes": 4.0 * h * w * (S * Cc + Cc + S * D + S)}
    if x3 and uses_split():
        check(_call(f"warpcorr_views_c{Cc}", work, _lib.lib().effi_warpcorr_views_x3_f32, _p(ref_nhwc), _ptr_array(srcs_nhwc), S,
                    _p(rt), _p(depth), dds, dps, Cc, h, w, D, _p(sim), _p(ent), int(_PRECISION == "bf16"), _stream()),
              "effi_warpcorr_views_x3_f32")
        return sim, ent
    check(_call(f"warpcorr_views_c{Cc}", work, _lib.lib().effi_warpcorr_views_f32, _p(ref_nhwc), _ptr_array(srcs_nhwc), S,
                _p(rt), _p(depth), dds, dps, Cc, h, w, D, _p(sim), _p(ent), _stream()), "effi_warpcorr_views_f32")
    return sim, ent


def warpcorr_views_bwd(ref_nhwc, srcs_nhwc, rt, depth, D, grad_sim):
    """Backward of ``warpcorr_views``' similarity output (scope row n2): grad_sim [S,D,h,w] -> (grad_ref [h,w,C], [grad_src_v [h,w,C]])."""
    h, w, Cc = ref_nhwc.shape
    S = len(srcs_nhwc)
    _t(ref_nhwc, "ref_nhwc"), _t(rt, "rt"), _t(depth, "depth", contiguous=False), _t(grad_sim, "grad_sim")
    for s_ in srcs_nhwc:
        _t(s_, "src_nhwc")
    if tuple(grad_sim.shape) != (S, D, h, w):
        raise ValueError(f"grad_sim {tuple(grad_sim.shape)} does not match (S, D, h, w) = {(S, D, h, w)}")
    depth, dds, dps = _depth_strides(depth, D, h, w)
    g_ref = torch.zeros_like(ref_nhwc)          # accumulated per source view by the windowed kernel
    g_src = [torch.zeros_like(s_) for s_ in srcs_nhwc]
    check(_lib.lib().effi_warpcorr_views_bwd_f32(_p(ref_nhwc), _ptr_array(srcs_nhwc), S, _p(rt), _p(depth), dds, dps, Cc, h, w, D,
                                                  _p(grad_sim), _p(g_ref), _ptr_array(g_src), _stream()), "effi_warpcorr_views_bwd_f32")
    return g_ref, g_src


VIEW_TABLE_ROW = MAX_VIEWS + 2        # include/effi_mvs_hip.h: EFFI_VIEW_TABLE_ROW


class ViewTable:
    """The feature maps of one reference view's (reference, sources) item as a table of device pointers the warp kernels read
    WHEN THEY RUN (``effi_warpcorr_views_tbl_f32`` / ``effi_warpcorr_dyn_tbl_f32``): ``ptrs`` int64 [n_stages, VIEW_TABLE_ROW],
    row s = channel-last maps [h_s, w_s, C_s] of the reference (entry 0) and of the sources (entries 1..S).  A captured graph of
    the hot path keeps the table's address, not the maps': ``set()`` points it at another item's cached maps with one small
    launch on the current stream (scan_eval.py) -- no 150-MB copy into static inputs.  ``shapes``: per stage (C, h, w).

    The maps are NOT kept alive by the table: whoever calls ``set`` owns them until every replay that reads them has finished
    (scan_eval.ScanFeatureCache does, per scan)."""

    def __init__(self, shapes, n_views, device, ptrs=None):
        if not 2 <= n_views <= MAX_VIEWS + 1 or not 1 <= len(shapes) <= 4:
            raise ValueError("ViewTable: 2..MAX_VIEWS+1 views, 1..4 stages")
        self.shapes = [tuple(int(v) for v in sh) for sh in shapes]
        self.n_views = int(n_views)
        self.ptrs = torch.zeros(len(shapes), VIEW_TABLE_ROW, dtype=torch.int64, device=device) if ptrs is None else ptrs
        if tuple(self.ptrs.shape) != (len(shapes), VIEW_TABLE_ROW) or self.ptrs.dtype != torch.int64 or not self.ptrs.is_contiguous():
            raise ValueError("ViewTable: ptrs must be a contiguous int64 [n_stages, VIEW_TABLE_ROW] tensor")

    def rebind(self, ptrs):
        """The same geometry over another pointer tensor (a graph slot's static copy)."""
        return ViewTable(self.shapes, self.n_views, ptrs.device, ptrs)

    def set(self, maps):
        """maps[s][v]: channel-last [h_s, w_s, C_s] fp32 map of view v (0 = reference) at stage s."""
        vals = []
        for s_, (sh, row) in enumerate(zip(self.shapes, maps)):
            if len(row) != self.n_views:
                raise ValueError(f"ViewTable.set: stage {s_} has {len(row)} maps, the table was built for {self.n_views} views")
            for m in row:
                _t(m, "feature map")
                if tuple(m.shape) != (sh[1], sh[2], sh[0]):
                    raise ValueError(f"ViewTable.set: stage {s_} map {tuple(m.shape)} is not channel-last {(sh[1], sh[2], sh[0])}")
            vals += [m.data_ptr() for m in row] + [0] * (VIEW_TABLE_ROW - len(row))
        arr = (C.c_void_p * len(vals))(*vals)
        check(_lib.lib().effi_view_table_set(C.c_void_p(self.ptrs.data_ptr()), arr, len(vals), _stream()), "effi_view_table_set")

    def row(self, s_):
        return C.c_void_p(self.ptrs.data_ptr() + 8 * VIEW_TABLE_ROW * s_)


def warpcorr_views_tbl(table, stage, rt, depth, D, x3=False):
    """``warpcorr_views`` reading its reference / source maps through row ``stage`` of a ViewTable."""
    Cc, h, w = table.shapes[stage]
    S = table.n_views - 1
    _t(rt, "rt"), _t(depth, "depth", contiguous=False)
    if rt.shape[0] != S:
        raise ValueError("Different number of images and projection matrices")
    depth, dds, dps = _depth_strides(depth, D, h, w)
    sim = torch.empty(S, D, h, w, device=rt.device, dtype=torch.float32)
    ent = torch.empty(S, h, w, device=rt.device, dtype=torch.float32)
    work = lambda: {"flops": S * D * h * w * (10.0 * Cc + 20), "bytes": 4.0 * h * w * (S * Cc + Cc + S * D + S)}
    if x3 and uses_split():
        check(_call(f"warpcorr_views_c{Cc}", work, _lib.lib().effi_warpcorr_views_x3_tbl_f32, table.row(stage), S, _p(rt), _p(depth), dds,
                    dps, Cc, h, w, D, _p(sim), _p(ent), int(_PRECISION == "bf16"), _stream()), "effi_warpcorr_views_x3_tbl_f32")
        return sim, ent
    check(_call(f"warpcorr_views_c{Cc}", work, _lib.lib().effi_warpcorr_views_tbl_f32, table.row(stage), S, _p(rt), _p(depth), dds, dps,
                Cc, h, w, D, _p(sim), _p(ent), _stream()), "effi_warpcorr_views_tbl_f32")
    return sim, ent


def warpcorr_dyn_tbl(table, stage, rt, cur_depth, interval, view_w, D):
    """``warpcorr_dyn`` reading its reference / source maps through row ``stage`` of a ViewTable."""
    Cc, h, w = table.shapes[stage]
    S = table.n_views - 1
    _t(rt, "rt"), _t(cur_depth, "cur_depth"), _t(interval, "interval"), _t(view_w, "view_w")
    if view_w.shape[0] != S or rt.shape[0] != S:
        raise ValueError("view weights / projections / sources disagree on the number of views")
    vh, vw = view_w.shape[1], view_w.shape[2]
    shift = 0
    while (vh << shift) < h:
        shift += 1
    if (vh << shift) != h or (vw << shift) != w:
        raise ValueError(f"view weights {vh}x{vw} are not a power-of-two downsampling of {h}x{w}")
    sim = torch.empty(D, h, w, device=rt.device, dtype=torch.float32)
    samples = torch.empty(D, h, w, device=rt.device, dtype=torch.float32)
    work = lambda: {"flops": S * D * h * w * (10.0 * Cc + 20), "bytes": 4.0 * h * w * (S * Cc + Cc + 2 * D + 1 + S / 4.0 ** shift)}
    check(_call(f"warpcorr_dyn_c{Cc}", work, _lib.lib().effi_warpcorr_dyn_tbl_f32, table.row(stage), S, _p(rt), _p(cur_depth),
                _p(interval), _p(view_w), shift, Cc, h, w, D, _p(sim), _p(samples), _stream()), "effi_warpcorr_dyn_tbl_f32")
    return sim, samples


def pixelwise_net(entropy, params):
    n, h, w = entropy.shape
    _t(entropy, "entropy"), _t(params, "params")
    out = torch.empty_like(entropy)
    check(_lib.lib().effi_pixelwise_net_f32(_p(entropy), _p(params), n, h, w, _p(out), _stream()),
          "effi_pixelwise_net_f32")
    return out


def view_aggregate(sim_views, weights):
    """sum_v sim_v w_v / (sum_v w_v + 1e-6); ``weights`` None = the plain mean over the views (pixel_wise_net = None)."""
    S, D, h, w = sim_views.shape
    _t(sim_views, "sim_views")
    if weights is not None:
        _t(weights, "weights")
    out = torch.empty(D, h, w, device=sim_views.device, dtype=torch.float32)
    check(_lib.lib().effi_view_aggregate_f32(_p(sim_views), _p(weights), S, D, h * w, _p(out), _stream()),
          "effi_view_aggregate_f32")
    return out


def warpcorr_dyn(ref_nhwc, srcs_nhwc, rt, cur_depth, interval, view_w, D):
    """cur_depth [h,w]; interval: 1-element tensor; view_w [S,h>>k,w>>k] -> (sim [D,h,w], samples [D,h,w])."""
    h, w, Cc = ref_nhwc.shape
    S = len(srcs_nhwc)
    _t(ref_nhwc, "ref_nhwc"), _t(rt, "rt"), _t(cur_depth, "cur_depth"), _t(interval, "interval"), _t(view_w, "view_w")
    for s in srcs_nhwc:
        _t(s, "src_nhwc")
    if view_w.shape[0] != S or rt.shape[0] != S:
        raise ValueError("view weights / projections / sources disagree on the number of views")
    vh, vw = view_w.shape[1], view_w.shape[2]
    shift = 0
    while (vh << shift) < h:
        shift += 1
    if (vh << shift) != h or (vw << shift) != w:
        raise ValueError(f"view weights {vh}x{vw} are not a power-of-two downsampling of {h}x{w}")
    sim = torch.empty(D, h, w, device=ref_nhwc.device, dtype=torch.float32)
    samples = torch.empty(D, h, w, device=ref_nhwc.device, dtype=torch.float32)
    work = lambda: {"flops": S * D * h * w * (10.0 * Cc + 20), "bytes": 4.0 * h * w * (S * Cc + Cc + 2 * D + 1 + S / 4.0 ** shift)}
    check(_call(f"warpcorr_dyn_c{Cc}", work, _lib.lib().effi_warpcorr_dyn_f32, _p(ref_nhwc), _ptr_array(srcs_nhwc), S, _p(rt),
                _p(cur_depth), _p(interval), _p(view_w), shift, Cc, h, w, D, _p(sim), _p(samples), _stream()),
          "effi_warpcorr_dyn_f32")
    return sim, samples


def conv3d_k3(srcs, weight, bias, cout, stride=(1, 1, 1), relu=True, skip=None):
    """srcs: list of planar [Ci,D,h,w]; weight packed [cin,27,cout]; -> [cout,Do,ho,wo]."""
    for s in srcs:
        _t(s, "conv3d input")
    _, D, h, w = srcs[0].shape
    sz, sxy = int(stride[0]), int(stride[1])
    Do, ho, wo = (D - 1) // sz + 1, (h - 1) // sxy + 1, (w - 1) // sxy + 1
    out = torch.empty(cout, Do, ho, wo, device=srcs[0].device, dtype=torch.float32)
    if skip is not None:
        _t(skip, "skip")
        assert skip.shape == out.shape
    cin = sum(s.shape[0] for s in srcs)
    work = lambda: {"flops": 2.0 * 27 * cin * cout * Do * ho * wo,
                    "bytes": 4.0 * (cin * D * h * w + cout * Do * ho * wo * (2 if skip is not None else 1))}
    check(_call(f"conv3d_c{'8' if cout % 8 == 0 else '1'}_s{sz}{sxy}", work, _lib.lib().effi_conv3d_k3_f32, _ptr_array(srcs),
                _int_array([s.shape[0] for s in srcs]), len(srcs), _p(weight), _p(bias), cout, D, h, w, sz, sxy, int(relu),
                _p(skip), _p(out), _stream()), "effi_conv3d_k3_f32")
    return out


def conv3d_k3s1_mfma(x, wpack, bias, cout, relu=True):
    """x planar [cin,D,h,w]; stride-1 3-D conv as z-batched 2-D MFMA convs -> [cout,D,h,w]."""
    _t(x, "conv3d input")
    cin, D, h, w = x.shape
    out = torch.empty(cout, D, h, w, device=x.device, dtype=torch.float32)
    work = lambda: {"flops": 2.0 * 27 * cin * cout * D * h * w, "bytes": 4.0 * (cin + cout) * D * h * w}
    check(_call(f"conv3d_mfma_nt{cout // 16}", work, _lib.lib().effi_conv3d_k3s1_mfma_f32, _p(x), cin, _p(wpack), _p(bias), cout,
                D, h, w, int(relu), _p(out), _stream()), "effi_conv3d_k3s1_mfma_f32")
    return out


def conv3d_k3s1_bf16x3(srcs, wpack, bias, cout, relu=True):
    """srcs: planar [Ci,D,h,w] tensors (channel concatenation); stride-1 3-D conv as z-batched 2-D convs in split precision
    (``packing.pack_conv3d_planes_bf16x3``) -> [cout,D,h,w].  w % 4 == 0, cout <= 32."""
    for s in srcs:
        _t(s, "conv3d input")
    _, D, h, w = srcs[0].shape
    cin = sum(s.shape[0] for s in srcs)
    out = torch.empty(cout, D, h, w, device=srcs[0].device, dtype=torch.float32)
    work = lambda: {"flops": 2.0 * 27 * cin * cout * D * h * w, "bytes": 4.0 * (cin + cout) * D * h * w}
    check(_call(f"conv3d_x3_nt{(cout + 15) // 16}", work, _x3("effi_conv3d_k3s1_bf16x3_f32"), _ptr_array(srcs),
                _int_array([s.shape[0] for s in srcs]), len(srcs), _p(wpack), _p(bias), cout, D, h, w, int(relu), _p(out),
                _stream()), "effi_conv3d_k3s1_bf16x3_f32")
    return out


def conv3d_k3s1_roll(srcs, wpack, bias, cout, relu=True):
    """srcs: one or two planar [Ci,D,h,w] tensors, 8 or 16 channels in total; stride-1 3-D conv with a rolling window of
    input planes in split precision (``packing.pack_conv3d_roll_bf16x3``) -> [cout,D,h,w].  w % 4 == 0, cout <= 32."""
    for s in srcs:
        _t(s, "conv3d input")
    _, D, h, w = srcs[0].shape
    cin = sum(s.shape[0] for s in srcs)
    out = torch.empty(cout, D, h, w, device=srcs[0].device, dtype=torch.float32)
    work = lambda: {"flops": 2.0 * 27 * cin * cout * D * h * w, "bytes": 4.0 * (cin + cout) * D * h * w}
    check(_call(f"conv3d_roll_oct{cin // 8}_nt{(cout + 15) // 16}", work, _x3("effi_conv3d_k3s1_roll_bf16x3_f32"), _ptr_array(srcs),
                _int_array([s.shape[0] for s in srcs]), len(srcs), _p(wpack), _p(bias), cout, D, h, w, int(relu), _p(out),
                _stream()), "effi_conv3d_k3s1_roll_bf16x3_f32")
    return out


def conv3d_k3s2_x3(x, wpack, bias, cout, relu=True):
    """x planar [cin,D,h,w] (w % 4 == 0); stride-(2,2,2) 3-D conv in split precision (``packing.pack_conv3d_s2_bf16x3``) ->
    [cout,Do,ho,wo]."""
    _t(x, "conv3d input")
    cin, D, h, w = x.shape
    Do, ho, wo = (D - 1) // 2 + 1, (h - 1) // 2 + 1, (w - 1) // 2 + 1
    out = torch.empty(cout, Do, ho, wo, device=x.device, dtype=torch.float32)
    work = lambda: {"flops": 2.0 * 27 * cin * cout * Do * ho * wo, "bytes": 4.0 * (cin * D * h * w + cout * Do * ho * wo)}
    check(_call(f"conv3d_s2x3_nt{(cout + 15) // 16}", work, _x3("effi_conv3d_k3s2_bf16x3_f32"), _p(x), cin, _p(wpack), _p(bias), cout,
                D, h, w, int(relu), _p(out), _stream()), "effi_conv3d_k3s2_bf16x3_f32")
    return out


def conv3d_k3s2_mfma(x, wpack, bias, cout, relu=True):
    """x planar [cin,D,h,w]; stride-(2,2,2) 3-D conv as z-batched stride-2 2-D MFMA convs -> [cout,Do,ho,wo]."""
    _t(x, "conv3d input")
    cin, D, h, w = x.shape
    Do, ho, wo = (D - 1) // 2 + 1, (h - 1) // 2 + 1, (w - 1) // 2 + 1
    out = torch.empty(cout, Do, ho, wo, device=x.device, dtype=torch.float32)
    work = lambda: {"flops": 2.0 * 27 * cin * cout * Do * ho * wo, "bytes": 4.0 * (cin * D * h * w + cout * Do * ho * wo)}
    check(_call(f"conv3d_mfma_s2_nt{cout // 16}", work, _lib.lib().effi_conv3d_k3s2_mfma_f32, _p(x), cin, _p(wpack), _p(bias), cout,
                D, h, w, int(relu), _p(out), _stream()), "effi_conv3d_k3s2_mfma_f32")
    return out


def deconv3d_k3(x, weight, bias, cout, sz=2, relu=True, skip=None):
    _t(x, "deconv3d input")
    cin, D, h, w = x.shape
    out = torch.empty(cout, sz * D, 2 * h, 2 * w, device=x.device, dtype=torch.float32)
    if skip is not None:
        _t(skip, "skip")
        assert skip.shape == out.shape, f"skip {tuple(skip.shape)} vs out {tuple(out.shape)}"
    work = lambda: {"flops": 2.0 * 27 * cin * cout * D * h * w,
                    "bytes": 4.0 * (cin * D * h * w + cout * sz * D * 4 * h * w * (2 if skip is not None else 1))}
    check(_call(f"deconv3d_c{cout if cout == 1 else 8}_s{sz}", work, _lib.lib().effi_deconv3d_k3_f32, _p(x), cin, _p(weight),
                _p(bias), cout, D, h, w, sz, int(relu), _p(skip), _p(out), _stream()), "effi_deconv3d_k3_f32")
    return out


def deconv3d_k3s2_x3(x, wpack, bias, cout, relu=True, skip=None):
    """Transposed 3-D conv, stride (2,2,2), on the bf16 matrix cores in split precision
    (``packing.pack_deconv3d_s2_bf16x3``): x [cin,D,h,w] -> [cout,2D,2h,2w] (+ skip after the ReLU)."""
    _t(x, "deconv3d input")
    cin, D, h, w = x.shape
    out = torch.empty(cout, 2 * D, 2 * h, 2 * w, device=x.device, dtype=torch.float32)
    if skip is not None:
        _t(skip, "skip")
        assert skip.shape == out.shape, f"skip {tuple(skip.shape)} vs out {tuple(out.shape)}"
    work = lambda: {"flops": 2.0 * 27 * cin * cout * D * h * w,
                    "bytes": 4.0 * (cin * D * h * w + cout * 8 * D * h * w * (2 if skip is not None else 1))}
    check(_call("deconv3d_x3", work, _x3("effi_deconv3d_k3s2_bf16x3_f32"), _p(x), cin, _p(wpack), _p(bias), cout,
                D, h, w, int(relu), _p(skip), _p(out), _stream()), "effi_deconv3d_k3s2_bf16x3_f32")
    return out


def fusion_dynamic_filter(ref_depth, src_depths, ref_cam, src_cams, ref_conf=None, prob_threshold=0.0, dh_view_num=2,
                          dist_base=4.0, rel_diff_base=1300.0, relative=False, want_points=True, want_reproj=False):
    """Scope row n3: one reference view through the dynamic geometric-consistency filter (misc/fusion.py:117-181,
    test_tank.py:466-512).  ref_depth [h,w]; src_depths [V,h,w]; ref_cam [2,4,4]; src_cams [V,2,4,4]; ref_conf [H,W] or None
    -> dict(depth [h,w], geo_mask / prob_mask / mask [h,w] uint8, points [3,h,w] or None, reproj_xyd [V,3,h,w] or None)."""
    for name, t_ in (("ref_depth", ref_depth), ("src_depths", src_depths), ("ref_cam", ref_cam), ("src_cams", src_cams)):
        _t(t_, name)
    V, h, w = src_depths.shape
    dev = ref_depth.device
    if tuple(ref_depth.shape) != (h, w) or tuple(ref_cam.shape) != (2, 4, 4) or tuple(src_cams.shape) != (V, 2, 4, 4):
        raise ValueError("fusion_dynamic_filter: ref_depth [h,w], src_depths [V,h,w], ref_cam [2,4,4], src_cams [V,2,4,4]")
    ch = cw = 0
    if ref_conf is not None:
        _t(ref_conf, "ref_conf")
        ch, cw = ref_conf.shape
    out = {"depth": torch.empty(h, w, device=dev, dtype=torch.float32),
           "geo_mask": torch.empty(h, w, device=dev, dtype=torch.uint8),
           "prob_mask": torch.empty(h, w, device=dev, dtype=torch.uint8),
           "mask": torch.empty(h, w, device=dev, dtype=torch.uint8),
           "points": torch.empty(3, h, w, device=dev, dtype=torch.float32) if want_points else None,
           "reproj_xyd": torch.empty(V, 3, h, w, device=dev, dtype=torch.float32) if want_reproj else None}
    scratch = torch.empty(52 * (V + 1), device=dev, dtype=torch.float32)
    work = lambda: {"flops": 0.0, "bytes": 4.0 * h * w * (1 + V + 1 + (3 if want_points else 0)) + 3.0 * h * w}
    check(_call("fusion_dynamic_filter", work, _lib.lib().effi_fusion_dynamic_filter_f32, _p(ref_depth), _p(src_depths),
                _p(ref_cam), _p(src_cams), V, h, w, _p(ref_conf), ch, cw, float(prob_threshold), int(dh_view_num),
                float(dist_base), float(rel_diff_base), int(bool(relative)), _p(scratch), _p(out["depth"]), _p(out["geo_mask"]),
                _p(out["prob_mask"]), _p(out["mask"]), _p(out["points"]), _p(out["reproj_xyd"]), _stream()),
          "effi_fusion_dynamic_filter_f32")
    return out


def fusion_vis_filter(ref_depth, reproj_xyd, dist_base=4.0, rel_diff_base=1300.0, thres_view=2, relative=False):
    """misc/fusion.py:157-181 (vis_filter_dynamic's arithmetic): ref_depth [n,1,h,w], reproj_xyd [n,v,3,h,w] ->
    masks [n,v,v+1-thres_view,h,w] uint8."""
    _t(ref_depth, "ref_depth"), _t(reproj_xyd, "reproj_xyd")
    n, v, three, h, w = reproj_xyd.shape
    if three != 3 or tuple(ref_depth.shape) != (n, 1, h, w):
        raise ValueError("fusion_vis_filter: ref_depth [n,1,h,w], reproj_xyd [n,v,3,h,w]")
    nthr = v + 1 - int(thres_view)
    masks = torch.empty(n, v, max(nthr, 0), h, w, device=ref_depth.device, dtype=torch.uint8)
    if nthr <= 0:
        return masks
    check(_lib.lib().effi_fusion_vis_filter_f32(_p(ref_depth), _p(reproj_xyd), n, v, h, w, float(dist_base), float(rel_diff_base),
                                                int(thres_view), int(bool(relative)), _p(masks), _stream()), "effi_fusion_vis_filter_f32")
    return masks


def fusion_dtu_reproject(ref_depth, src_depth, ref_cam, src_cam, s=None, e=None, dist_base=0.5, diff_base=0.25):
    """test_dtu_dypcd.py:164-233 for one (reference, source) pair (PARITY UNPINNED): depth maps [h,w], cameras [2,4,4] ->
    (out5 [5,h,w] = depth_reprojected, x_reprojected, y_reprojected, x_src, y_src; masks [e-s,h,w] uint8 or None).  With s / e
    given the call is check_geometric_consistency (masks + zeroing under the last mask), without it reproject_with_depth."""
    for name, t_ in (("ref_depth", ref_depth), ("src_depth", src_depth), ("ref_cam", ref_cam), ("src_cam", src_cam)):
        _t(t_, name)
    h, w = ref_depth.shape
    if tuple(src_depth.shape) != (h, w) or tuple(ref_cam.shape) != (2, 4, 4) or tuple(src_cam.shape) != (2, 4, 4):
        raise ValueError("fusion_dtu_reproject: depth maps [h,w], cameras [2,4,4]")
    dev = ref_depth.device
    out5 = torch.empty(5, h, w, device=dev, dtype=torch.float32)
    masks = torch.empty(e - s, h, w, device=dev, dtype=torch.uint8) if s is not None else None
    scratch = torch.empty(104, device=dev, dtype=torch.float32)
    check(_lib.lib().effi_fusion_dtu_reproject_f32(_p(ref_depth), _p(src_depth), _p(ref_cam), _p(src_cam), h, w, int(s or 1), int(e or 2),
                                                   float(dist_base), float(diff_base), _p(scratch), _p(out5), _p(masks), _stream()),
          "effi_fusion_dtu_reproject_f32")
    return out5, masks


FUSION_IMG2CAM, FUSION_CAM2WORLD, FUSION_WORLD2CAM, FUSION_CAM2IMG = 0, 1, 2, 3


def fusion_points(mode, pts, cam, depth=None):
    """The point transforms of misc/fusion.py:23-47 with one camera per batch element.  pts [n or 1,h,w,3 or 4,1] (a leading 1 is
    broadcast over the batch, as the reference's ``@`` does), cam [n,2,4,4], depth [n,1,h,w] for FUSION_IMG2CAM -> [n,h,w,4 or 3,1]."""
    _t(pts, "points"), _t(cam, "cam")
    n = cam.shape[0]
    ni = 3 if mode == FUSION_IMG2CAM else 4
    no = 3 if mode == FUSION_CAM2IMG else 4
    if pts.dim() != 5 or pts.shape[3] != ni or pts.shape[4] != 1 or pts.shape[0] not in (1, n) or tuple(cam.shape[1:]) != (2, 4, 4):
        raise ValueError(f"fusion_points: points [n|1,h,w,{ni},1] and cam [n,2,4,4] expected, got {tuple(pts.shape)} / {tuple(cam.shape)}")
    h, w = pts.shape[1], pts.shape[2]
    if mode == FUSION_IMG2CAM:
        _t(depth, "depth")
        if tuple(depth.shape) != (n, 1, h, w):
            raise ValueError(f"fusion_points: depth {tuple(depth.shape)} is not [n,1,h,w] = {(n, 1, h, w)}")
    out = torch.empty(n, h, w, no, 1, device=pts.device, dtype=torch.float32)
    scratch = torch.empty(52 * n, device=pts.device, dtype=torch.float32)
    bstride = 0 if (pts.shape[0] == 1 and n > 1) else h * w * ni
    check(_lib.lib().effi_fusion_points_f32(int(mode), _p(pts), bstride, _p(depth), _p(cam), n, h, w, _p(scratch), _p(out), _stream()),
          "effi_fusion_points_f32")
    return out


def fusion_dtu_filter(ref_depth, src_depths, ref_cam, src_cams, confidence=None, conf_threshold=0.5, conf_keep=0.75, s=1, e=11,
                      dist_base=0.5, diff_base=0.25, want_points=True):
    """Scope row n3, DTU branch (test_dtu_dypcd.py:164-333; PARITY UNPINNED, see the header): one reference view through the dynamic
    geometric-consistency filter.  ref_depth [h,w]; src_depths [V,h,w]; ref_cam [2,4,4]; src_cams [V,2,4,4]; confidence [ch,cw] (any
    size: resized to the depth size as cv2.resize does) or None -> dict(depth [h,w], photo_mask / geo_mask / mask [h,w] uint8,
    points [3,h,w] or None)."""
    for name, t_ in (("ref_depth", ref_depth), ("src_depths", src_depths), ("ref_cam", ref_cam), ("src_cams", src_cams)):
        _t(t_, name)
    V, h, w = src_depths.shape
    dev = ref_depth.device
    if tuple(ref_depth.shape) != (h, w) or tuple(ref_cam.shape) != (2, 4, 4) or tuple(src_cams.shape) != (V, 2, 4, 4):
        raise ValueError("fusion_dtu_filter: ref_depth [h,w], src_depths [V,h,w], ref_cam [2,4,4], src_cams [V,2,4,4]")
    conf = None
    if confidence is not None:
        _t(confidence, "confidence")
        conf = confidence if tuple(confidence.shape) == (h, w) else resize_planar(confidence.reshape(1, *confidence.shape[-2:]).contiguous(), h, w)[0]
    out = {"depth": torch.empty(h, w, device=dev, dtype=torch.float32),
           "photo_mask": torch.empty(h, w, device=dev, dtype=torch.uint8),
           "geo_mask": torch.empty(h, w, device=dev, dtype=torch.uint8),
           "mask": torch.empty(h, w, device=dev, dtype=torch.uint8),
           "points": torch.empty(3, h, w, device=dev, dtype=torch.float32) if want_points else None}
    scratch = torch.empty(52 * (V + 1), device=dev, dtype=torch.float32)
    work = lambda: {"flops": 0.0, "bytes": 4.0 * h * w * (1 + V + 1 + 1 + (3 if want_points else 0)) + 3.0 * h * w}
    check(_call("fusion_dtu_filter", work, _lib.lib().effi_fusion_dtu_filter_f32, _p(ref_depth), _p(src_depths), _p(ref_cam), _p(src_cams),
                V, h, w, _p(conf), float(conf_threshold), float(conf_keep), int(s), int(e), float(dist_base), float(diff_base), _p(scratch),
                _p(out["depth"]), _p(out["photo_mask"]), _p(out["geo_mask"]), _p(out["mask"]), _p(out["points"]), _stream()),
          "effi_fusion_dtu_filter_f32")
    return out


def image_prepare(img_u8, dst_h, dst_w, out=None):
    """Scope row n4: decoded image [h,w,3] (or [h,w]) uint8 on the device -> [3,dst_h,dst_w] fp32 = cv2-style bilinear resize of
    img / 255, channel-first (datasets/general_eval.py:83-117,189)."""
    if not isinstance(img_u8, torch.Tensor) or img_u8.dtype != torch.uint8 or not img_u8.is_contiguous():
        raise TypeError("image_prepare: contiguous uint8 tensor [h,w,c] expected")
    if img_u8.dim() == 2:
        img_u8 = img_u8.unsqueeze(-1)
    if not img_u8.is_cuda:
        raise _lib.EffiLibraryError("effi_image_prepare_u8_f32 needs a tensor on the GPU (there is no CPU fallback)")
    h, w, c = img_u8.shape
    if out is None:
        out = torch.empty(c, dst_h, dst_w, device=img_u8.device, dtype=torch.float32)
    work = lambda: {"flops": 0.0, "bytes": 1.0 * h * w * c + 4.0 * c * dst_h * dst_w}
    check(_call("image_prepare", work, _lib.lib().effi_image_prepare_u8_f32, C.c_void_p(img_u8.data_ptr()), h, w, c, dst_h, dst_w,
                _p(out), _stream()), "effi_image_prepare_u8_f32")
    return out


def resize_planar(x, dst_h, dst_w, out=None):
    """[C,h,w] fp32 -> [C,dst_h,dst_w], the bilinear resize of ``image_prepare`` (datasets/general_eval.py:160-166)."""
    _t(x, "image")
    c, h, w = x.shape
    if out is None:
        out = torch.empty(c, dst_h, dst_w, device=x.device, dtype=torch.float32)
    check(_lib.lib().effi_resize_linear_f32(_p(x), c, h, w, dst_h, dst_w, _p(out), _stream()), "effi_resize_linear_f32")
    return out


def softmax_regress_conf(logits, depth, disp_range=None):
    """logits [D,h,w]; depth [D] / [D,h,w] -> (depth [h,w], confidence [h,w]) and, with ``disp_range``, also the regressed
    depth as normalised inverse depth (``depth_to_inv`` of it, written by the same kernel)."""
    D, h, w = logits.shape
    _t(logits, "logits"), _t(depth, "depth", contiguous=False)
    depth, dds, dps = _depth_strides(depth, D, h, w)
    od = torch.empty(h, w, device=logits.device, dtype=torch.float32)
    oc = torch.empty(h, w, device=logits.device, dtype=torch.float32)
    oi, n_range = None, 0
    if disp_range is not None:
        _t(disp_range, "disp_range")
        oi, n_range = torch.empty(h, w, device=logits.device, dtype=torch.float32), disp_range.numel()
    check(_lib.lib().effi_softmax_regress_conf_f32(_p(logits), _p(depth), dds, dps, D, h * w, _p(od), _p(oc), _p(disp_range), n_range,
                                                    _p(oi), _stream()), "effi_softmax_regress_conf_f32")
    return (od, oc) if disp_range is None else (od, oc, oi)


def _vol_strides(vol, h, w):
    """Per-pixel D-vector volume as planar [D,h,w] or pixel-major [h*w,1,1,D] (the reference's `pro`)."""
    _t(vol, "volume", contiguous=False)
    if vol.dim() == 3:
        if not vol.is_contiguous():
            vol = vol.contiguous()
        return vol, h * w, 1, vol.shape[0]
    if vol.dim() == 4 and vol.shape[0] == h * w:
        return vol, vol.stride(3), vol.stride(0), vol.shape[3]
    raise ValueError(f"unsupported volume shape {tuple(vol.shape)}")


def _range_ptr(r, h, w):
    """depth-range bound: scalar-like tensor (global) or per-pixel [.., h, w] map."""
    _t(r, "range", contiguous=False)
    if r.numel() == 1:
        return r.reshape(1), 0
    if r.numel() == h * w:
        return r.reshape(h, w).contiguous(), 1
    raise ValueError(f"depth range with {r.numel()} elements does not match {h}x{w}")


def vol_lookup1d(vol, query, dmin, dmax, h, w):
    """query [nq, h', w'] with (h', w') == (h, w) or exactly 2x (read nearest-downsampled)."""
    vol, vds, vps, Dp = _vol_strides(vol, h, w)
    _t(query, "query")
    nq, qh, qw = query.shape
    if (qh, qw) == (h, w):
        qys, qxs = qw, 1
    elif (qh // 2, qw // 2) == (h, w):
        qys, qxs = 2 * qw, 2        # F.interpolate(nearest) to half size picks the even samples
    else:
        raise ValueError("query resolution must equal the volume's or be twice it")
    dmin_t, rps = _range_ptr(dmin, h, w)
    dmax_t, rps2 = _range_ptr(dmax, h, w)
    if rps != rps2:
        raise ValueError("depth_min / depth_max must both be global or both per-pixel")
    out = torch.empty(nq, h, w, device=query.device, dtype=torch.float32)
    check(_lib.lib().effi_vol_lookup1d_f32(_p(vol), vds, vps, Dp, _p(query), qh * qw, qys, qxs, nq, _p(dmin_t),
                                            _p(dmax_t), rps, h, w, _p(out), _stream()), "effi_vol_lookup1d_f32")
    return out


def bilinear_sampler1d(img, coords, want_mask=False):
    """``bilinear_sampler`` of the reference (models/Effi_MVS_plus.py:102-117) for H == 1 images: img [N,C,1,W], coords
    [N,Ho,Wo,2] pixel (x, y) -> [N,C,Ho,Wo] (and the in-range mask [N,Ho,Wo,1] as float)."""
    _t(img, "img"), _t(coords, "coords")
    N, Cc, H, W = img.shape
    if H != 1 or coords.dim() != 4 or coords.shape[0] != N or coords.shape[-1] != 2:
        raise ValueError("bilinear_sampler1d: img [N,C,1,W] and coords [N,Ho,Wo,2]")
    Ho, Wo = coords.shape[1:3]
    out = torch.empty(N, Cc, Ho, Wo, device=img.device, dtype=torch.float32)
    mask = torch.empty(N, Ho, Wo, 1, device=img.device, dtype=torch.float32) if want_mask else None
    check(_lib.lib().effi_bilinear_sampler1d_f32(_p(img), N, Cc, W, _p(coords), Ho * Wo, _p(out), _p(mask), _stream()),
          "effi_bilinear_sampler1d_f32")
    return (out, mask) if want_mask else out


def vol_lookup1d_pair(vol_a, vol_b, query, dmin, dmax, h, w):
    """``vol_lookup1d`` into two planar volumes of the same shape with the same queries, one launch -> (out_a, out_b)."""
    vol_a, vds, vps, Dp = _vol_strides(vol_a, h, w)
    vol_b, vds_b, vps_b, Dp_b = _vol_strides(vol_b, h, w)
    if (vds, vps, Dp) != (vds_b, vps_b, Dp_b):
        raise ValueError("vol_lookup1d_pair: the two volumes must share shape and layout")
    _t(query, "query")
    nq, qh, qw = query.shape
    if (qh, qw) == (h, w):
        qys, qxs = qw, 1
    elif (qh // 2, qw // 2) == (h, w):
        qys, qxs = 2 * qw, 2
    else:
        raise ValueError("query resolution must equal the volume's or be twice it")
    dmin_t, rps = _range_ptr(dmin, h, w)
    dmax_t, rps2 = _range_ptr(dmax, h, w)
    if rps != rps2:
        raise ValueError("depth_min / depth_max must both be global or both per-pixel")
    out_a = torch.empty(nq, h, w, device=query.device, dtype=torch.float32)
    out_b = torch.empty_like(out_a)
    check(_lib.lib().effi_vol_lookup1d_pair_f32(_p(vol_a), _p(vol_b), vds, vps, Dp, _p(query), qh * qw, qys, qxs, nq, _p(dmin_t),
                                                 _p(dmax_t), rps, h, w, _p(out_a), _p(out_b), _stream()), "effi_vol_lookup1d_pair_f32")
    return out_a, out_b


def conv3d_k3_pair(x_a, weight_a, bias_a, x_b, weight_b, bias_b, cout, sxy=1, relu=True):
    """``conv3d_k3`` of two single-source convolutions of the same shape (cout 8, stride (1,sxy,sxy)) in one launch."""
    _t(x_a, "conv3d input"), _t(x_b, "conv3d input")
    if x_a.shape != x_b.shape:
        raise ValueError("conv3d_k3_pair: the two inputs must share their shape")
    cin, D, h, w = x_a.shape
    ho, wo = (h - 1) // sxy + 1, (w - 1) // sxy + 1
    out_a = torch.empty(cout, D, ho, wo, device=x_a.device, dtype=torch.float32)
    out_b = torch.empty_like(out_a)
    work = lambda: {"flops": 2.0 * 2 * 27 * cin * cout * D * ho * wo, "bytes": 2 * 4.0 * (cin * D * h * w + cout * D * ho * wo)}
    check(_call(f"conv3d_pair_c8_s1{sxy}", work, _lib.lib().effi_conv3d_k3_pair_f32, _p(x_a), _p(weight_a), _p(bias_a), _p(out_a),
                _p(x_b), _p(weight_b), _p(bias_b), _p(out_b), cin, cout, D, h, w, sxy, int(relu), _stream()), "effi_conv3d_k3_pair_f32")
    return out_a, out_b


def csp_gen_roll_pair(x, prior_a, w0_a, b0_a, wc_a, bc_a, w1_a, b1_a, prior_b, w0_b, b0_b, wc_b, bc_b, w1_b, b1_b):
    """conv0 | conv_cost -> conv1 of two cross-scale blocks over one fine volume in ONE launch (``effi_csp_gen_roll_bf16x3_pair_f32``):
    x [1,D,H,W]; priors [1,D,h,w]; w0 / wc [1,27,8] packed (``Conv3d._packed``), w1 the rolling operand -> (c1_a, c1_b) [8,D,h,w].
    Bitwise ``conv3d_k3_pair`` (sxy 2 / sxy 1) + ``conv3d_k3s1_roll_pair``."""
    _t(x, "fine volume"), _t(prior_a, "prior"), _t(prior_b, "prior")
    _, D, H, W = x.shape
    h, w = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    if tuple(prior_a.shape) != (1, D, h, w) or tuple(prior_b.shape) != (1, D, h, w):
        raise ValueError(f"csp_gen_roll_pair: priors must be [1,{D},{h},{w}], got {tuple(prior_a.shape)} / {tuple(prior_b.shape)}")
    for t_, n_ in ((w0_a, 216), (wc_a, 216), (w0_b, 216), (wc_b, 216), (b0_a, 8), (bc_a, 8), (b0_b, 8), (bc_b, 8)):
        _t(t_, "generated layer's parameters")
        if t_.numel() != n_:
            raise ValueError("csp_gen_roll_pair: the generated layers are 1 -> 8 channels (27 x 8 weights, 8 biases)")
    out_a = torch.empty(8, D, h, w, device=x.device, dtype=torch.float32)
    out_b = torch.empty_like(out_a)
    V = D * h * w
    work = lambda: {"flops": 2 * 2.0 * 27 * (8 * 2 + 16 * 8) * V, "bytes": 4.0 * (D * H * W + 2 * V + 2 * 8 * V)}
    check(_call("csp_gen_roll_pair", work, _x3("effi_csp_gen_roll_bf16x3_pair_f32"), _p(x), D, H, W, _p(prior_a), _p(w0_a), _p(b0_a),
                _p(wc_a), _p(bc_a), _p(w1_a), _p(b1_a), _p(out_a), _p(prior_b), _p(w0_b), _p(b0_b), _p(wc_b), _p(bc_b), _p(w1_b),
                _p(b1_b), _p(out_b), _stream()), "effi_csp_gen_roll_bf16x3_pair_f32")
    return out_a, out_b


def conv3d_k3s1_roll_pair(srcs_a, wpack_a, bias_a, srcs_b, wpack_b, bias_b, cout, relu=True):
    """``conv3d_k3s1_roll`` of two convolutions with the same source split and shape in one launch."""
    for s in list(srcs_a) + list(srcs_b):
        _t(s, "conv3d input")
    if [tuple(s.shape) for s in srcs_a] != [tuple(s.shape) for s in srcs_b]:
        raise ValueError("conv3d_k3s1_roll_pair: the two source lists must share their shapes")
    _, D, h, w = srcs_a[0].shape
    cin = sum(s.shape[0] for s in srcs_a)
    out_a = torch.empty(cout, D, h, w, device=srcs_a[0].device, dtype=torch.float32)
    out_b = torch.empty_like(out_a)
    work = lambda: {"flops": 2 * 2.0 * 27 * cin * cout * D * h * w, "bytes": 2 * 4.0 * (cin + cout) * D * h * w}
    check(_call(f"conv3d_roll_pair_oct{cin // 8}", work, _x3("effi_conv3d_k3s1_roll_bf16x3_pair_f32"), _ptr_array(srcs_a),
                _p(wpack_a), _p(bias_a), _p(out_a), _ptr_array(srcs_b), _p(wpack_b), _p(bias_b), _p(out_b),
                _int_array([s.shape[0] for s in srcs_a]), len(srcs_a), cout, D, h, w, int(relu), _stream()),
          "effi_conv3d_k3s1_roll_bf16x3_pair_f32")
    return out_a, out_b


def deconv3d_k3_pair(x_a, weight_a, bias_a, x_b, weight_b, bias_b, cout, sz=1, relu=True):
    """``deconv3d_k3`` (stride (1,2,2), cout 1) of two inputs of the same shape in one launch."""
    _t(x_a, "deconv3d input"), _t(x_b, "deconv3d input")
    if x_a.shape != x_b.shape:
        raise ValueError("deconv3d_k3_pair: the two inputs must share their shape")
    cin, D, h, w = x_a.shape
    out_a = torch.empty(cout, sz * D, 2 * h, 2 * w, device=x_a.device, dtype=torch.float32)
    out_b = torch.empty_like(out_a)
    work = lambda: {"flops": 2 * 2.0 * 27 * cin * cout * D * h * w, "bytes": 2 * 4.0 * (cin * D * h * w + cout * sz * D * 4 * h * w)}
    check(_call("deconv3d_pair_c1_s1", work, _lib.lib().effi_deconv3d_k3_pair_f32, _p(x_a), _p(weight_a), _p(bias_a), _p(out_a),
                _p(x_b), _p(weight_b), _p(bias_b), _p(out_b), cin, cout, D, h, w, sz, int(relu), _stream()), "effi_deconv3d_k3_pair_f32")
    return out_a, out_b


def getcost(x, disp_range, interval, cur_vol, reg_vol, dmin, dmax, nq, h, w, input_is_depth=False, out=None):
    _t(x, "inv_depth"), _t(interval, "interval")
    cur_vol, cds, cps, Dc = _vol_strides(cur_vol, h, w)
    reg_vol, rds, rps, Dr = _vol_strides(reg_vol, h, w)
    dmin_t, gps = _range_ptr(dmin, h, w)
    dmax_t, gps2 = _range_ptr(dmax, h, w)
    if gps != gps2:
        raise ValueError("depth_min / depth_max must both be global or both per-pixel")
    if out is None:
        out = torch.empty(2 * nq, h, w, device=x.device, dtype=torch.float32)
    n_range = 0 if disp_range is None else disp_range.numel()
    check(_lib.lib().effi_getcost_f32(_p(x), _p(disp_range), n_range, int(input_is_depth), _p(interval),
                                       _p(cur_vol), cds, cps, Dc, _p(reg_vol), rds, rps, Dr, _p(dmin_t), _p(dmax_t), gps,
                                       nq, h, w, _p(out), _stream()), "effi_getcost_f32")
    return out


def getcost_conv1x1(x, disp_range, interval, cur_vol, reg_vol, dmin, dmax, nq, h, w, weight, bias, cout, relu=True,
                    input_is_depth=False, out=None):
    """``getcost`` + 1x1 convolution (weight [2*nq, cout], bias [cout]) + ReLU in one kernel -> [cout,h,w]."""
    _t(x, "inv_depth"), _t(interval, "interval"), _t(weight, "weight"), _t(bias, "bias")
    cur_vol, cds, cps, Dc = _vol_strides(cur_vol, h, w)
    reg_vol, rds, rps, Dr = _vol_strides(reg_vol, h, w)
    dmin_t, gps = _range_ptr(dmin, h, w)
    dmax_t, gps2 = _range_ptr(dmax, h, w)
    if gps != gps2:
        raise ValueError("depth_min / depth_max must both be global or both per-pixel")
    if out is None:
        out = torch.empty(cout, h, w, device=x.device, dtype=torch.float32)
    n_range = 0 if disp_range is None else disp_range.numel()
    work = lambda: {"flops": 2.0 * h * w * 2 * nq * cout, "bytes": 4.0 * h * w * (cout + 1 + Dc + Dr)}
    check(_call("getcost_conv1x1", work, _lib.lib().effi_getcost_conv1x1_f32, _p(x), _p(disp_range), n_range,
                int(input_is_depth), _p(interval), _p(cur_vol), cds, cps, Dc, _p(reg_vol), rds, rps, Dr, _p(dmin_t), _p(dmax_t),
                gps, nq, h, w, _p(weight), _p(bias), cout, int(relu), _p(out), _stream()), "effi_getcost_conv1x1_f32")
    return out


def encoder_inputs(x, disp_range, interval, cur_vol, reg_vol, dmin, dmax, nq, h, w, weight_c1, bias_c1, weight_d1, bias_d1, cout,
                   out_c1=None, out_d1=None):
    """``getcost_conv1x1`` (+ReLU) and ``conv2d_c1k7_relu`` of the same normalised inverse-depth map in one launch ->
    (relu(convc1(cost)) [cout,h,w], relu(convd1(x)) [cout,h,w]).  nq == 3, cout in {16, 32, 48}."""
    _t(x, "inv_depth"), _t(interval, "interval"), _t(disp_range, "disp_range")
    for t_ in (weight_c1, bias_c1, weight_d1, bias_d1):
        _t(t_, "encoder weights")
    cur_vol, cds, cps, Dc = _vol_strides(cur_vol, h, w)
    reg_vol, rds, rps, Dr = _vol_strides(reg_vol, h, w)
    dmin_t, gps = _range_ptr(dmin, h, w)
    dmax_t, gps2 = _range_ptr(dmax, h, w)
    if gps != gps2:
        raise ValueError("depth_min / depth_max must both be global or both per-pixel")
    if out_c1 is None:
        out_c1 = torch.empty(cout, h, w, device=x.device, dtype=torch.float32)
    if out_d1 is None:
        out_d1 = torch.empty(cout, h, w, device=x.device, dtype=torch.float32)
    work = lambda: {"flops": 2.0 * h * w * (2 * nq + 49) * cout, "bytes": 4.0 * h * w * (2 * cout + 1 + Dc + Dr)}
    # split / bf16 precision: the 7x7 half on the matrix cores (EFFI_C1K7_MFMA=0: the exact-fp32 vector form, as "fp32" precision uses)
    x3 = uses_split() and _PY_OPTS["c1k7_mfma"] != 0
    fn = _lib.lib().effi_encoder_inputs_bf16x3_f32 if x3 else _lib.lib().effi_encoder_inputs_f32
    check(_call("encoder_inputs", work, fn, _p(x), _p(disp_range), disp_range.numel(),
                _p(interval), _p(cur_vol), cds, cps, Dc, _p(reg_vol), rds, rps, Dr, _p(dmin_t), _p(dmax_t), gps, nq, h, w,
                _p(weight_c1), _p(bias_c1), _p(weight_d1), _p(bias_d1), cout, _p(out_c1), _p(out_d1), _stream()),
          "effi_encoder_inputs_bf16x3_f32" if x3 else "effi_encoder_inputs_f32")
    return out_c1, out_d1


def conv2d_k3_bf16x3_pair(srcs_a, wpack_a, bias_a, srcs_b, wpack_b, bias_b, cout, act=ACT_NONE, out_a=None, out_b=None):
    """Two independent split-precision 3x3 convolutions of the same shape in one launch (``conv2d_k3_bf16x3`` twice)."""
    for s in list(srcs_a) + list(srcs_b):
        _t(s, "conv2d input")
    h, w = srcs_a[0].shape[-2:]
    if tuple(srcs_b[0].shape[-2:]) != (h, w):
        raise ValueError("conv2d_k3_bf16x3_pair: the two convolutions must share the map size")
    dev = srcs_a[0].device
    if out_a is None:
        out_a = torch.empty(cout, h, w, device=dev, dtype=torch.float32)
    if out_b is None:
        out_b = torch.empty(cout, h, w, device=dev, dtype=torch.float32)
    cin = sum(s.shape[0] for s in srcs_a) + sum(s.shape[0] for s in srcs_b)
    work = lambda: {"flops": 2.0 * h * w * cin * cout * 9, "bytes": 4.0 * h * w * (cin + 2 * cout)}
    check(_call(f"conv2d_k3x3_pair_nt{(cout + 15) // 16}", work, _x3("effi_conv2d_k3_bf16x3_pair_f32"), _ptr_array(srcs_a),
                _int_array([s.shape[0] for s in srcs_a]), len(srcs_a), _p(wpack_a), _p(bias_a), _p(out_a), _ptr_array(srcs_b),
                _int_array([s.shape[0] for s in srcs_b]), len(srcs_b), _p(wpack_b), _p(bias_b), _p(out_b), cout, h, w, act,
                _stream()), "effi_conv2d_k3_bf16x3_pair_f32")
    return out_a, out_b


def conv2d(srcs, wpack, bias, cout, ks, epilogue=EPI_PLAIN, act=ACT_NONE, aux0=None, aux1=None, disp_range=None,
           out0=None, out1=None):
    for s in srcs:
        _t(s, "conv2d input")
    h, w = srcs[0].shape[-2:]
    dev = srcs[0].device
    if out0 is None:
        if epilogue == EPI_GRU_ZR:
            out0 = torch.empty(cout // 2, h, w, device=dev, dtype=torch.float32)
        elif epilogue == EPI_HEAD:
            out0 = torch.empty(1, h, w, device=dev, dtype=torch.float32)
        elif epilogue == EPI_NHWC:
            out0 = torch.empty(h, w, cout, device=dev, dtype=torch.float32)
        else:
            out0 = torch.empty(cout, h, w, device=dev, dtype=torch.float32)
    if out1 is None and epilogue == EPI_GRU_ZR:
        out1 = torch.empty(cout // 2, h, w, device=dev, dtype=torch.float32)
    if out1 is None and epilogue == EPI_HEAD:
        out1 = torch.empty(1, h, w, device=dev, dtype=torch.float32)
    n_range = 0 if disp_range is None else disp_range.numel()
    cin = sum(s.shape[0] for s in srcs)
    if hasattr(wpack, "w32"):                 # packing.Conv2dWeights: both operand orders, pick the arithmetic here
        if (uses_split() and wpack.wx is not None and ks == 3 and w % 4 == 0
                and epilogue in (EPI_PLAIN, EPI_NHWC, EPI_GRU_ZR, EPI_GRU_Q)
                and all(s.shape[0] % 8 == 0 for s in srcs[:-1])):
            return conv2d_k3_bf16x3(srcs, wpack.wx, bias, cout, epilogue=epilogue, act=act, aux0=aux0, aux1=aux1,
                                    out0=out0, out1=out1)
        wpack = wpack.w32
    work = lambda: {"flops": 2.0 * h * w * cin * cout * ks * ks, "bytes": 4.0 * h * w * (cin + cout)}
    check(_call(f"conv2d_k{ks}_nt{(cout + 15) // 16}_epi{epilogue}", work, _lib.lib().effi_conv2d_f32, _ptr_array(srcs),
                _int_array([s.shape[0] for s in srcs]), len(srcs), _p(wpack), _p(bias), cout, ks, h, w, epilogue, act,
                _p(aux0), _p(aux1), _p(disp_range), n_range, _p(out0), _p(out1), _stream()), "effi_conv2d_f32")
    return (out0, out1) if out1 is not None else out0


def conv2d_k3_bf16x3(srcs, wpack, bias, cout, epilogue=EPI_PLAIN, act=ACT_NONE, aux0=None, aux1=None, out0=None, out1=None):
    """3x3 convolution in split precision (hi*hi + hi*lo + lo*hi on the bf16 MFMA, fp32 accumulate); ``wpack`` from
    ``packing.pack_conv2d_bf16x3``.  Same sources / epilogues as ``conv2d`` (PLAIN, NHWC, GRU_ZR, GRU_Q); w % 4 == 0 and every
    source but the last has a multiple of 8 channels (otherwise the library reports UNSUPPORTED)."""
    for s in srcs:
        _t(s, "conv2d input")
    h, w = srcs[0].shape[-2:]
    dev = srcs[0].device
    if out0 is None:
        if epilogue == EPI_GRU_ZR:
            out0 = torch.empty(cout // 2, h, w, device=dev, dtype=torch.float32)
        elif epilogue in (EPI_NHWC, EPI_NHWC_ADD_SHUF2):
            out0 = torch.empty(h, w, cout, device=dev, dtype=torch.float32)
        else:
            out0 = torch.empty(cout, h, w, device=dev, dtype=torch.float32)
    if out1 is None and epilogue == EPI_GRU_ZR:
        out1 = torch.empty(cout // 2, h, w, device=dev, dtype=torch.float32)
    if epilogue in (EPI_ADD_SHUF2, EPI_NHWC_ADD_SHUF2):
        _t(aux0, "coarser map")
        if tuple(aux0.shape) != (4 * cout, h // 2, w // 2) or h % 2 or w % 2:
            raise ValueError("conv2d_k3_bf16x3: the pixel-shuffled map must be [4*cout, h/2, w/2]")
    cin = sum(s.shape[0] for s in srcs)
    work = lambda: {"flops": 2.0 * h * w * cin * cout * 9, "bytes": 4.0 * h * w * (cin + cout)}
    check(_call(f"conv2d_k3x3_nt{(cout + 15) // 16}_epi{epilogue}", work, _x3("effi_conv2d_k3_bf16x3_f32"), _ptr_array(srcs),
                _int_array([s.shape[0] for s in srcs]), len(srcs), _p(wpack), _p(bias), cout, h, w, epilogue, act,
                _p(aux0), _p(aux1), None, 0, _p(out0), _p(out1), _stream()), "effi_conv2d_k3_bf16x3_f32")
    return (out0, out1) if out1 is not None else out0


# =============================================================================================
# split-resident ("SR") maps of the update block: include/effi_mvs_hip.h, section "Split-resident activation maps"
# =============================================================================================
_SR = os.environ.get("EFFI_MVS_SR", "1") != "0"


def set_sr(enabled):
    """Route the update block's split-precision convolutions through split-resident maps (default) or through fp32 maps (the
    round-2 form; bitwise the same results)."""
    global _SR
    _SR = bool(enabled)


def uses_sr(pixels=None):
    """SR maps on?  ``pixels``: size of the map in question -- EFFI_MVS_SR_MAX_PX (A/B runs) keeps larger maps on the fp32 form."""
    if not (_SR and uses_split()):
        return False
    return pixels is None or pixels <= _SR_MAX_PX


_SR_MAX_PX = int(os.environ.get("EFFI_MVS_SR_MAX_PX", str(1 << 40)))


def sr_geometry(h, w):
    """(hp, wp) of an SR plane for an h x w map: effi_sr_geometry."""
    return ((h + 15) // 16) * 16 + 2, (((w + 63) // 64) * 64 if w >= 512 else ((w + 15) // 16) * 16) + 2


class SRMap:
    """One split-resident map: ``t`` is a bf16 tensor [channels/8, 2 (hi, lo), hp, wp, 8]; pixel (y, x) sits at [.., y+1, x+1, :]."""
    __slots__ = ("t", "channels", "h", "w", "hp", "wp")

    def __init__(self, t, channels, h, w):
        self.t, self.channels, self.h, self.w = t, channels, h, w
        self.hp, self.wp = t.shape[2], t.shape[3]

    def parts(self):
        """(hi, lo) as fp32 [channels, h, w] tensors (tests / debugging; plain torch)."""
        v = self.t[:, :, 1:self.h + 1, 1:self.w + 1, :].float()          # [o, 2, h, w, 8]
        v = v.permute(1, 0, 4, 2, 3).reshape(2, self.channels, self.h, self.w)
        return v[0], v[1]


def sr_alloc(n_maps, channels, h, w, device, clear=True):
    """``n_maps`` SR maps of one geometry in one allocation -> list of SRMap.  The border is zeroed here (one small launch) unless
    ``clear`` is False (the caller clears several blocks at once with ``sr_clear_border``)."""
    if channels % 8:
        raise ValueError("SR maps hold whole octets of channels")
    hp, wp = sr_geometry(h, w)
    block = torch.empty(n_maps, channels // 8, 2, hp, wp, 8, device=device, dtype=torch.bfloat16)
    maps = [SRMap(block[i], channels, h, w) for i in range(n_maps)]
    if clear:
        sr_clear_border([maps])
    return maps


def _sr(m, name):
    if not isinstance(m, SRMap):
        raise TypeError(f"{name}: expected an SRMap")
    t_ = m.t
    if not t_.is_cuda:
        raise EffiLibraryError(f"{name}: CPU tensor passed to the HIP path (no CPU fallback exists)")
    if t_.device.index != torch._C._cuda_getDevice():
        raise EffiLibraryError(f"{name}: map lives on cuda:{t_.device.index} but the current device is cuda:{torch._C._cuda_getDevice()}")
    if t_.dtype != torch.bfloat16 or not t_.is_contiguous():
        raise ValueError(f"{name}: SR maps are contiguous bf16 tensors")
    return m


def sr_clear_border(groups):
    """Zero the borders of up to 4 groups of SR maps (each group: consecutive maps of ONE allocation and geometry) in one launch."""
    if not 1 <= len(groups) <= 4:
        raise ValueError("sr_clear_border: 1..4 groups")
    firsts = [_sr(g[0], "SR map") for g in groups]
    planes = [2 * (g[0].channels // 8) * len(g) for g in groups]
    for g in groups:       # consecutive maps of one allocation
        step = g[0].t.numel() * 2
        if any(m.t.data_ptr() != g[0].t.data_ptr() + i * step for i, m in enumerate(g)):
            raise ValueError("sr_clear_border: a group must be consecutive maps of one sr_alloc block")
    check(_lib.lib().effi_sr_clear_border(_ptr_array([m.t for m in firsts]), _int_array(planes), _int_array([m.h for m in firsts]),
                                           _int_array([m.w for m in firsts]), _int_array([m.hp for m in firsts]),
                                           _int_array([m.wp for m in firsts]), len(groups), _stream()), "effi_sr_clear_border")


def sr_from_planar(x, out=None):
    """fp32 [C,h,w] -> SRMap (a block boundary of the update block: e.g. a caller-supplied hidden state)."""
    _t(x, "map")
    Cc, h, w = x.shape
    if out is None:
        out = sr_alloc(1, Cc, h, w, x.device)[0]
    check(_lib.lib().effi_sr_from_planar_f32(_p(x), Cc, h, w, _p(_sr(out, "SR map").t), out.hp, out.wp, _stream()), "effi_sr_from_planar_f32")
    return out


def split_tanh_relu_stages_sr(ctxs, hds, cds, hidden_srs, q4=None):
    """``split_tanh_relu_stages`` that also writes each hidden state into the given SRMap -> [(hidden, inp), ...] (fp32).
    ``q4``: per stage, write the fp32 hidden state in the Q4 layout (same shape [hd,h,w] tensor, values ordered [hd/4][h][w][4])."""
    outs = []
    for c_, hd, cd, m in zip(ctxs, hds, cds, hidden_srs):
        _t(c_, "context"), _sr(m, "hidden SR map")
        _, h, w = c_.shape
        if (m.channels, m.h, m.w) != (hd, h, w):
            raise ValueError("split_tanh_relu_stages_sr: SR map does not match the hidden state")
        outs.append((torch.empty(hd, h, w, device=c_.device, dtype=torch.float32),
                     torch.empty(cd, h, w, device=c_.device, dtype=torch.float32)))
    if len(ctxs) > 4:
        raise ValueError("split_tanh_relu_stages_sr: up to 4 stages")
    check(_lib.lib().effi_split_tanh_relu_stages_sr_f32(
        _ptr_array(ctxs), _int_array(list(hds)), _int_array(list(cds)), _int_array([c_.shape[1] for c_ in ctxs]),
        _int_array([c_.shape[2] for c_ in ctxs]), _ptr_array([o[0] for o in outs]), _ptr_array([m.t for m in hidden_srs]),
        _int_array([m.hp for m in hidden_srs]), _int_array([m.wp for m in hidden_srs]), _ptr_array([o[1] for o in outs]),
        _int_array([int(bool(v)) for v in (q4 if q4 is not None else [0] * len(ctxs))]), len(ctxs),
        _stream()), "effi_split_tanh_relu_stages_sr_f32")
    return outs


def encoder_inputs_sr(x, disp_range, interval, cur_vol, reg_vol, dmin, dmax, nq, h, w, weight_c1, bias_c1, weight_d1, bias_d1, cout,
                      out_c1, out_d1):
    """``encoder_inputs`` (split precision) writing relu(convc1(cost)) and relu(convd1(x)) as SR maps."""
    _t(x, "inv_depth"), _t(interval, "interval"), _t(disp_range, "disp_range")
    for t_ in (weight_c1, bias_c1, weight_d1, bias_d1):
        _t(t_, "encoder weights")
    _sr(out_c1, "out_c1"), _sr(out_d1, "out_d1")
    cur_vol, cds, cps, Dc = _vol_strides(cur_vol, h, w)
    reg_vol, rds, rps, Dr = _vol_strides(reg_vol, h, w)
    dmin_t, gps = _range_ptr(dmin, h, w)
    dmax_t, gps2 = _range_ptr(dmax, h, w)
    if gps != gps2:
        raise ValueError("depth_min / depth_max must both be global or both per-pixel")
    if (out_c1.channels, out_c1.h, out_c1.w) != (cout, h, w) or (out_d1.channels, out_d1.h, out_d1.w) != (cout, h, w):
        raise ValueError("encoder_inputs_sr: output maps must be [cout,h,w]")
    work = lambda: {"flops": 2.0 * h * w * (2 * nq + 49) * cout, "bytes": 4.0 * h * w * (2 * cout + 1 + Dc + Dr)}
    check(_call("encoder_inputs", work, _lib.lib().effi_encoder_inputs_bf16x3_sr, _p(x), _p(disp_range), disp_range.numel(),
                _p(interval), _p(cur_vol), cds, cps, Dc, _p(reg_vol), rds, rps, Dr, _p(dmin_t), _p(dmax_t), gps, nq, h, w,
                _p(weight_c1), _p(bias_c1), _p(weight_d1), _p(bias_d1), cout, _p(out_c1.t), _p(out_d1.t), out_c1.hp, out_c1.wp,
                _stream()), "effi_encoder_inputs_bf16x3_sr")
    return out_c1, out_d1


def encoder_pair_gen_sr(x, disp_range, interval, cur_vol, reg_vol, dmin, dmax, nq, h, w, weight_c1, bias_c1, weight_d1, bias_d1, hd,
                        wpack_c2, bias_c2, out_c2, wpack_d2, bias_d2, out_d2, cout, act=ACT_RELU):
    """``encoder_inputs_sr`` + ``conv2d_k3_pair_sr`` in one launch (models/update.py:86-91): the 1x1 / 7x7 results are generated tile by
    tile inside the 3x3 kernel.  -> (act(convc2(cor1)), act(convd2(dfm1))) as SR maps; bitwise equal to the two launches."""
    _t(x, "inv_depth"), _t(interval, "interval"), _t(disp_range, "disp_range")
    for t_ in (weight_c1, bias_c1, weight_d1, bias_d1):
        _t(t_, "encoder weights")
    g = _sr_srcs([out_c2, out_d2])
    cur_vol, cds, cps, Dc = _vol_strides(cur_vol, h, w)
    reg_vol, rds, rps, Dr = _vol_strides(reg_vol, h, w)
    dmin_t, gps = _range_ptr(dmin, h, w)
    dmax_t, gps2 = _range_ptr(dmax, h, w)
    if gps != gps2:
        raise ValueError("depth_min / depth_max must both be global or both per-pixel")
    if (g.channels, g.h, g.w) != (cout, h, w) or tuple(x.shape[-2:]) != (h, w):
        raise ValueError("encoder_pair_gen_sr: output maps must be [cout,h,w]")
    work = lambda: {"flops": 2.0 * h * w * ((2 * nq + 49) * hd + 2 * 9 * hd * cout), "bytes": 4.0 * h * w * (2 * cout + 1 + Dc + Dr)}
    check(_call(f"encgen_pair_nt{(cout + 15) // 16}", work, _x3("effi_encoder_pair_gen_bf16x3_sr"), _p(x), _p(disp_range),
                disp_range.numel(), _p(interval), _p(cur_vol), cds, cps, Dc, _p(reg_vol), rds, rps, Dr, _p(dmin_t), _p(dmax_t), gps, nq,
                h, w, _p(weight_c1), _p(bias_c1), _p(weight_d1), _p(bias_d1), hd, _p(wpack_c2), _p(bias_c2), _p(out_c2.t), _p(wpack_d2),
                _p(bias_d2), _p(out_d2.t), cout, g.hp, g.wp, act, _stream()), "effi_encoder_pair_gen_bf16x3_sr")
    return out_c2, out_d2


def _sr_srcs(srcs):
    for m in srcs:
        _sr(m, "conv2d input")
        if m.channels % 16:
            raise ValueError("SR convolution sources hold multiples of 16 channels")
    g = srcs[0]
    if any((m.h, m.w, m.hp, m.wp) != (g.h, g.w, g.hp, g.wp) for m in srcs):
        raise ValueError("SR convolution sources must share one geometry")
    return g


EPI_Q4 = 0x100          # include/effi_mvs_hip.h: EFFI_EPI_Q4


def q4_from_planar(x):
    """fp32 [C,h,w] -> the same values as [C/4,h,w,4] (the "Q4" layout of EFFI_EPI_Q4; tests and callers that hold planar state)."""
    C_, h, w = x.shape
    return x.view(C_ // 4, 4, h, w).permute(0, 2, 3, 1).contiguous()


def q4_to_planar(x, channels):
    """inverse of ``q4_from_planar``: a [C/4,h,w,4] block (any view of C*h*w floats) -> planar [C,h,w]."""
    h, w = x.shape[-3:-1] if x.dim() == 4 else (None, None)
    q = x.reshape(channels // 4, -1, 4) if h is None else x
    return q.permute(0, 3, 1, 2).reshape(channels, *q.shape[1:3]).contiguous()


def conv2d_k3_sr(srcs, wpack, bias, cout, epilogue=EPI_PLAIN, act=ACT_NONE, aux0=None, aux1=None, out0=None, out_sr=None, q4=False):
    """``conv2d_k3_bf16x3`` on SR maps.  PLAIN: -> out_sr (and out0 fp32 if given); GRU_ZR: -> (z fp32, r*h SR), aux0 = h fp32;
    GRU_Q: -> (h' fp32, h' SR), aux0 = h, aux1 = z.  ``q4`` (GRU epilogues): aux0 / aux1 / out0 hold their values as
    [C/4,h,w,4] instead of planar [C,h,w] (same number of floats; EFFI_EPI_Q4: one 16-byte access per lane and map)."""
    g = _sr_srcs(srcs)
    h, w = g.h, g.w
    dev = g.t.device
    c_out_sr = cout // 2 if epilogue == EPI_GRU_ZR else cout
    if out_sr is None:
        out_sr = sr_alloc(1, c_out_sr, h, w, dev)[0]
    _sr(out_sr, "out_sr")
    if (out_sr.channels, out_sr.h, out_sr.w) != (c_out_sr, h, w):
        raise ValueError("conv2d_k3_sr: output map does not match")
    if out0 is None and epilogue in (EPI_GRU_ZR, EPI_GRU_Q):
        out0 = torch.empty(c_out_sr, h, w, device=dev, dtype=torch.float32)
    for a_ in (aux0, aux1, out0):
        if a_ is not None:
            _t(a_, "fp32 map")
    cin = sum(m.channels for m in srcs)
    work = lambda: {"flops": 2.0 * h * w * cin * cout * 9, "bytes": 4.0 * h * w * (cin + cout)}
    check(_call(f"conv2d_k3x3_nt{(cout + 15) // 16}_epi{epilogue}", work, _x3("effi_conv2d_k3_bf16x3_sr"), _ptr_array([m.t for m in srcs]),
                _int_array([m.channels for m in srcs]), len(srcs), _p(wpack), _p(bias), cout, h, w, g.hp, g.wp,
                epilogue | (EPI_Q4 if (q4 and epilogue in (EPI_GRU_ZR, EPI_GRU_Q)) else 0), act,
                _p(aux0), _p(aux1), _p(out0), _p(out_sr.t), _stream()), "effi_conv2d_k3_bf16x3_sr")
    return (out0, out_sr) if out0 is not None else out_sr


def gru_zr_q_fused_sr(H_in, X, h_in, wzr_pack, bias_zr, wq_pack, bias_q, h_out, H_out):
    """ConvGRU in one launch (csrc/gru_fused.hpp): SR maps ``H_in`` (state) and ``X`` (encoder output), fp32 state ``h_in`` ->
    (``h_out`` fp32, ``H_out`` SR); outputs must be other buffers than the inputs.  Bitwise equal to ``conv2d_k3_sr(GRU_ZR)`` +
    ``conv2d_k3_sr(GRU_Q)``."""
    g = _sr_srcs([H_in, X, H_out])
    hd, h, w = g.channels, g.h, g.w
    _t(h_in, "h"), _t(h_out, "h_out")
    if tuple(h_in.shape) != (hd, h, w) or tuple(h_out.shape) != (hd, h, w) or X.channels != hd or H_out.channels != hd:
        raise ValueError("gru_zr_q_fused_sr: state / input maps must hold hd channels of one size")
    if h_out.data_ptr() == h_in.data_ptr() or H_out.t.data_ptr() in (H_in.t.data_ptr(), X.t.data_ptr()):
        raise ValueError("gru_zr_q_fused_sr: outputs must not alias inputs")
    work = lambda: {"flops": 2.0 * h * w * 9 * (2 * hd * 2 * hd + 2 * hd * hd), "bytes": 4.0 * h * w * 5 * hd}
    check(_call(f"gru_fused_nt{hd // 16}", work, _x3("effi_gru_zr_q_fused_bf16x3_sr"), _p(H_in.t), _p(X.t), _p(h_in), _p(wzr_pack), _p(bias_zr),
                _p(wq_pack), _p(bias_q), hd, h, w, g.hp, g.wp, _p(h_out), _p(H_out.t), _stream()), "effi_gru_zr_q_fused_bf16x3_sr")
    return h_out, H_out


def conv2d_k3_pair_sr(srcs_a, wpack_a, bias_a, out_a, srcs_b, wpack_b, bias_b, out_b, cout, act=ACT_NONE):
    """``conv2d_k3_bf16x3_pair`` on SR maps (SR in, SR out)."""
    g = _sr_srcs(list(srcs_a) + list(srcs_b) + [out_a, out_b])
    h, w = g.h, g.w
    cin = sum(m.channels for m in srcs_a) + sum(m.channels for m in srcs_b)
    work = lambda: {"flops": 2.0 * h * w * cin * cout * 9, "bytes": 4.0 * h * w * (cin + 2 * cout)}
    check(_call(f"conv2d_k3x3_pair_nt{(cout + 15) // 16}", work, _x3("effi_conv2d_k3_bf16x3_pair_sr"), _ptr_array([m.t for m in srcs_a]),
                _int_array([m.channels for m in srcs_a]), len(srcs_a), _p(wpack_a), _p(bias_a), _p(out_a.t),
                _ptr_array([m.t for m in srcs_b]), _int_array([m.channels for m in srcs_b]), len(srcs_b), _p(wpack_b), _p(bias_b),
                _p(out_b.t), cout, h, w, g.hp, g.wp, act, _stream()), "effi_conv2d_k3_bf16x3_pair_sr")
    return out_a, out_b


def conv2d_k3_k1_sr(srcs, wpack, bias, cout1, extra, w2pack, bias2, cout2, relu=True, relu1=False, out=None, out_sr=None):
    """``conv2d_k3_k1_x3`` on SR inputs -> ``out_sr`` (SRMap) if given, else fp32 [cout2,h,w]."""
    g = _sr_srcs(srcs)
    h, w = g.h, g.w
    c_extra = 0
    if extra is not None:
        _t(extra, "extra channels")
        c_extra = extra.shape[0]
    if out_sr is None and out is None:
        out = torch.empty(cout2, h, w, device=g.t.device, dtype=torch.float32)
    if out_sr is not None:
        _sr(out_sr, "out_sr")
        if (out_sr.channels, out_sr.h, out_sr.w) != (cout2, h, w):
            raise ValueError("conv2d_k3_k1_sr: output map does not match")
    cin = sum(m.channels for m in srcs)
    work = lambda: {"flops": 2.0 * h * w * (cin * cout1 * 9 + (cout1 + c_extra) * cout2),
                    "bytes": 4.0 * h * w * (cin + c_extra + cout2)}
    check(_call(f"conv2d_k3k1_nt{(cout1 + 15) // 16}", work, _x3("effi_conv2d_k3_k1_bf16x3_sr"), _ptr_array([m.t for m in srcs]),
                _int_array([m.channels for m in srcs]), len(srcs), _p(wpack), _p(bias), cout1, int(relu1), _p(extra), c_extra,
                _p(w2pack), _p(bias2), cout2, int(relu), h, w, g.hp, g.wp, _p(None if out_sr is not None else out),
                _p(out_sr.t if out_sr is not None else None), _stream()), "effi_conv2d_k3_k1_bf16x3_sr")
    return out_sr if out_sr is not None else out


def conv2d_k3_k1_up2x_sr(srcs, wpack, bias, cout1, w2pack, bias2, inv_depth, disp_range, want_depth_inv=True):
    """``conv2d_k3_k1_up2x`` on SR inputs."""
    g = _sr_srcs(srcs)
    _t(inv_depth, "inv_depth"), _t(disp_range, "disp_range")
    h, w = g.h, g.w
    dev = g.t.device
    out_depth = torch.empty(2 * h, 2 * w, device=dev, dtype=torch.float32)
    out_dinv = torch.empty(2 * h, 2 * w, device=dev, dtype=torch.float32) if want_depth_inv else None
    cin = sum(m.channels for m in srcs)
    work = lambda: {"flops": 2.0 * h * w * (cin * cout1 * 9 + cout1 * 36), "bytes": 4.0 * h * w * (cin + 1 + 8)}
    check(_call(f"conv2d_k3k1up_nt{(cout1 + 15) // 16}", work, _x3("effi_conv2d_k3_k1_up2x_bf16x3_sr"), _ptr_array([m.t for m in srcs]),
                _int_array([m.channels for m in srcs]), len(srcs), _p(wpack), _p(bias), cout1, _p(w2pack), _p(bias2), _p(inv_depth),
                _p(disp_range), disp_range.numel(), h, w, g.hp, g.wp, _p(out_depth), _p(out_dinv), _stream()),
          "effi_conv2d_k3_k1_up2x_bf16x3_sr")
    return out_depth, out_dinv


def conv2d_k3_k1_x3(srcs, wpack, bias, cout1, extra, w2pack, bias2, cout2, relu=True, out=None, relu1=False):
    """3x3 split-precision conv (ReLU if ``relu1``) + the 1x1 conv over cat(result, ``extra``) in one kernel
    (``packing.pack_conv2d_bf16x3`` / ``packing.pack_conv1x1_after``) -> [cout2,h,w]."""
    for s in srcs:
        _t(s, "conv2d input")
    h, w = srcs[0].shape[-2:]
    c_extra = 0
    if extra is not None:
        _t(extra, "extra channels")
        c_extra = extra.shape[0]
    if out is None:
        out = torch.empty(cout2, h, w, device=srcs[0].device, dtype=torch.float32)
    cin = sum(s.shape[0] for s in srcs)
    work = lambda: {"flops": 2.0 * h * w * (cin * cout1 * 9 + (cout1 + c_extra) * cout2),
                    "bytes": 4.0 * h * w * (cin + c_extra + cout2)}
    check(_call(f"conv2d_k3k1_nt{(cout1 + 15) // 16}", work, _x3("effi_conv2d_k3_k1_bf16x3_f32"), _ptr_array(srcs),
                _int_array([s.shape[0] for s in srcs]), len(srcs), _p(wpack), _p(bias), cout1, int(relu1), _p(extra), c_extra,
                _p(w2pack), _p(bias2), cout2, int(relu), h, w, _p(out), _stream()), "effi_conv2d_k3_k1_bf16x3_f32")
    return out


def conv2d_k3_twice(x, w1, b1, w2, b2, cout, out=None):
    """relu(conv3x3(relu(conv3x3(x)))) with at most 8 channels into each layer (8 between them) in one kernel, the intermediate map
    in LDS (``packing.pack_conv2d_bf16x3_oct`` weights): the pyramid's full-resolution block.  x [cin<=8,h,w] -> [cout<=8,h,w]."""
    _t(x, "conv input")
    cin, h, w = x.shape
    if out is None:
        out = torch.empty(cout, h, w, device=x.device, dtype=torch.float32)
    work = lambda: {"flops": 2.0 * h * w * 9 * (cin * 8 + 8 * cout), "bytes": 4.0 * h * w * (cin + cout)}
    check(_call("conv2d_k3_twice", work, _x3("effi_conv2d_k3_twice_bf16x3_f32"), _p(x), cin, _p(w1), _p(b1), _p(w2), _p(b2), cout, h, w,
                _p(out), _stream()), "effi_conv2d_k3_twice_bf16x3_f32")
    return out


def encoder_tail(cor1, dfm1, wc2, bc2, wd2, bd2, wd, bd, cmix, extra, w2pack, bias2, cout2, out=None):
    """relu(convc2(cor1)), relu(convd2(dfm1)) -> convd over their concatenation -> 1x1 convc over cat(., extra) + ReLU in ONE kernel
    (the two intermediate maps stay in LDS): ``conv2d_k3_bf16x3_pair`` followed by ``conv2d_k3_k1_x3``.  hd == 16 only."""
    _t(cor1, "cor1"), _t(dfm1, "dfm1"), _t(extra, "extra channels")
    hd, h, w = cor1.shape
    if dfm1.shape != cor1.shape or extra.shape[-2:] != cor1.shape[-2:]:
        raise ValueError("encoder_tail: cor1, dfm1 and the extra channels must share one map size")
    if out is None:
        out = torch.empty(cout2, h, w, device=cor1.device, dtype=torch.float32)
    c_extra = extra.shape[0]
    work = lambda: {"flops": 2.0 * h * w * (2 * hd * hd * 9 + 2 * hd * cmix * 9 + (cmix + c_extra) * cout2),
                    "bytes": 4.0 * h * w * (2 * hd + c_extra + cout2)}
    check(_call("encoder_tail", work, _x3("effi_encoder_tail_bf16x3_f32"), _p(cor1), _p(dfm1), hd, _p(wc2), _p(bc2), _p(wd2), _p(bd2),
                _p(wd), _p(bd), cmix, _p(extra), c_extra, _p(w2pack), _p(bias2), cout2, h, w, _p(out), _stream()),
          "effi_encoder_tail_bf16x3_f32")
    return out


def conv2d_k3_k1_up2x(srcs, wpack, bias, cout1, w2pack, bias2, inv_depth, disp_range, want_depth_inv=True):
    """Mask head (3x3 + ReLU + 1x1 to 36 channels) and the convex x2 upsampling it feeds in one kernel: inv_depth [1,h,w] or [h,w]
    -> (depth [2h,2w], depth_to_inv(depth) [2h,2w] or None).  cout1 in {32, 64, 96}."""
    for s_ in srcs:
        _t(s_, "conv2d input")
    _t(inv_depth, "inv_depth"), _t(disp_range, "disp_range")
    h, w = srcs[0].shape[-2:]
    dev = srcs[0].device
    out_depth = torch.empty(2 * h, 2 * w, device=dev, dtype=torch.float32)
    out_dinv = torch.empty(2 * h, 2 * w, device=dev, dtype=torch.float32) if want_depth_inv else None
    cin = sum(s_.shape[0] for s_ in srcs)
    work = lambda: {"flops": 2.0 * h * w * (cin * cout1 * 9 + cout1 * 36), "bytes": 4.0 * h * w * (cin + 1 + 8)}
    check(_call(f"conv2d_k3k1up_nt{(cout1 + 15) // 16}", work, _x3("effi_conv2d_k3_k1_up2x_bf16x3_f32"), _ptr_array(srcs),
                _int_array([s_.shape[0] for s_ in srcs]), len(srcs), _p(wpack), _p(bias), cout1, _p(w2pack), _p(bias2), _p(inv_depth),
                _p(disp_range), disp_range.numel(), h, w, _p(out_depth), _p(out_dinv), _stream()), "effi_conv2d_k3_k1_up2x_bf16x3_f32")
    return out_depth, out_dinv


def conv2d_k5s2(x, wpack, bias, cout, act=ACT_RELU):
    """5x5 stride-2 pad-2 convolution (+bias, activation): x planar [cin,hin,win] -> [cout,ceil(hin/2),ceil(win/2)]."""
    _t(x, "conv input")
    cin, hin, win = x.shape
    ho, wo = (hin - 1) // 2 + 1, (win - 1) // 2 + 1
    out = torch.empty(cout, ho, wo, device=x.device, dtype=torch.float32)
    work = lambda: {"flops": 2.0 * ho * wo * cin * cout * 25, "bytes": 4.0 * (hin * win * cin + ho * wo * cout)}
    if hasattr(wpack, "w32"):                 # packing.Conv2dWeights: pick the arithmetic here
        if uses_split() and wpack.wx is not None and win % 4 == 0 and _PY_OPTS["k5s2_split"] != 0:
            check(_call(f"conv2d_k5s2x3_nt{(cout + 15) // 16}", work, _x3("effi_conv2d_k5s2_bf16x3_f32"), _p(x), cin, _p(wpack.wx), _p(bias),
                        cout, hin, win, act, _p(out), _stream()), "effi_conv2d_k5s2_bf16x3_f32")
            return out
        wpack = wpack.w32
    check(_call(f"conv2d_k5s2_nt{(cout + 15) // 16}", work, _lib.lib().effi_conv2d_k5s2_f32, _p(x), cin, _p(wpack), _p(bias), cout,
                hin, win, act, _p(out), _stream()), "effi_conv2d_k5s2_f32")
    return out


def conv2d_c1k7_relu(x, weight, bias, cout, out=None, exact=False):
    """7x7 single-input-channel convolution + ReLU (ProjectionInput.convd1).  ``exact``: exact fp32 products whatever
    ``get_precision()`` says (the training path, whose weight gradient is taken against this output)."""
    _t(x, "conv7 input")
    h, w = x.shape[-2:]
    if out is None:
        out = torch.empty(cout, h, w, device=x.device, dtype=torch.float32)
    if not exact and uses_split() and _PY_OPTS["c1k7_mfma"] != 0:
        check(_lib.lib().effi_conv2d_c1k7_relu_bf16x3_f32(_p(x), _p(weight), _p(bias), cout, h, w, _p(out), _stream()),
              "effi_conv2d_c1k7_relu_bf16x3_f32")
        return out
    check(_lib.lib().effi_conv2d_c1k7_relu_f32(_p(x), _p(weight), _p(bias), cout, h, w, _p(out), _stream()),
          "effi_conv2d_c1k7_relu_f32")
    return out


def convex_upsample2x(inv_depth, mask, disp_range=None, want_inv=True, want_depth_inv=False):
    """-> (inv [2h,2w] or None, depth [2h,2w] or None); depth needs ``disp_range``.  ``want_depth_inv``: a third result,
    ``depth_to_inv(depth)`` (what the next stage starts from), written by the same kernel."""
    _t(inv_depth, "inv_depth"), _t(mask, "mask")
    h, w = inv_depth.shape[-2:]
    if mask.shape[0] != 36:
        raise NotImplementedError("convex upsampling is instantiated for ratio 2 (36 mask channels)")
    out_inv = torch.empty(2 * h, 2 * w, device=mask.device, dtype=torch.float32) if want_inv else None
    out_depth = None
    n_range = 0
    if disp_range is not None:
        _t(disp_range, "disp_range")
        n_range = disp_range.numel()
        out_depth = torch.empty(2 * h, 2 * w, device=mask.device, dtype=torch.float32)
    out_dinv = torch.empty(2 * h, 2 * w, device=mask.device, dtype=torch.float32) if (want_depth_inv and out_depth is not None) else None
    check(_lib.lib().effi_convex_upsample2x_f32(_p(inv_depth), _p(mask), _p(disp_range), n_range, h, w,
                                                 _p(out_inv), _p(out_depth), _p(out_dinv), _stream()), "effi_convex_upsample2x_f32")
    return (out_inv, out_depth, out_dinv) if want_depth_inv else (out_inv, out_depth)


def split_tanh_relu(ctx, hd, cd):
    _t(ctx, "context")
    _, h, w = ctx.shape
    hid = torch.empty(hd, h, w, device=ctx.device, dtype=torch.float32)
    inp = torch.empty(cd, h, w, device=ctx.device, dtype=torch.float32)
    check(_lib.lib().effi_split_tanh_relu_f32(_p(ctx), hd, cd, h * w, _p(hid), _p(inp), _stream()),
          "effi_split_tanh_relu_f32")
    return hid, inp


def split_tanh_relu_stages(ctxs, hds, cds):
    """``split_tanh_relu`` of up to 4 context maps (the stages of the cascade) in one launch -> [(hidden, inp), ...]."""
    outs, hws = [], []
    for c_, hd, cd in zip(ctxs, hds, cds):
        _t(c_, "context")
        _, h, w = c_.shape
        hws.append(h * w)
        outs.append((torch.empty(hd, h, w, device=c_.device, dtype=torch.float32),
                     torch.empty(cd, h, w, device=c_.device, dtype=torch.float32)))
    if len(ctxs) > 4:
        raise ValueError("split_tanh_relu_stages: up to 4 stages")
    check(_lib.lib().effi_split_tanh_relu_stages_f32(_ptr_array(ctxs), _int_array(list(hds)), _int_array(list(cds)), _int_array(hws),
                                                      _ptr_array([o[0] for o in outs]), _ptr_array([o[1] for o in outs]), len(ctxs),
                                                      _stream()), "effi_split_tanh_relu_stages_f32")
    return outs


def depth_to_inv(depth, disp_range):
    _t(depth, "depth"), _t(disp_range, "disp_range")
    out = torch.empty_like(depth)
    check(_lib.lib().effi_depth_to_inv_f32(_p(depth), _p(disp_range), disp_range.numel(), depth.numel(), _p(out),
                                            _stream()), "effi_depth_to_inv_f32")
    return out


def stage1_hypotheses(disp_range, D):
    _t(disp_range, "disp_range")
    depths = torch.empty(D, device=disp_range.device, dtype=torch.float32)
    intervals = torch.empty(5, device=disp_range.device, dtype=torch.float32)   # 3 intervals, depth_min_, depth_max_
    check(_lib.lib().effi_stage1_hypotheses_f32(_p(disp_range), disp_range.numel(), D, _p(depths), _p(intervals),
                                                 _stream()), "effi_stage1_hypotheses_f32")
    return depths, intervals


def upsample_nearest(x, f):
    _t(x, "map")
    Cc, h, w = x.shape
    out = torch.empty(Cc, h * f, w * f, device=x.device, dtype=torch.float32)
    check(_lib.lib().effi_upsample_nearest_f32(_p(x), Cc, h, w, f, _p(out), _stream()), "effi_upsample_nearest_f32")
    return out


def head_update(partial9, bias2, inv_depth, disp_range):
    """Tail of the depth head when conv2's nine tap projections were applied by conv1's kernel (``conv2d_k3_k1_x3`` with
    ``packing.pack_head_taps``): partial9 [9,h,w], inv_depth [1,h,w] -> (inv_depth + tanh(conv2), its depth), both [1,h,w]."""
    _t(partial9, "partial sums"), _t(bias2, "bias"), _t(inv_depth, "inv_depth"), _t(disp_range, "disp_range")
    if partial9.dim() != 3 or partial9.shape[0] != 9:
        raise ValueError("head_update: partial sums must be [9,h,w]")
    h, w = partial9.shape[-2:]
    if inv_depth.numel() != h * w:
        raise ValueError("head_update: inv_depth must hold h*w values")
    out_inv = torch.empty(1, h, w, device=partial9.device, dtype=torch.float32)
    out_depth = torch.empty(1, h, w, device=partial9.device, dtype=torch.float32)
    work = lambda: {"flops": 30.0 * h * w, "bytes": 4.0 * h * w * 12}
    check(_call("head_update", work, _lib.lib().effi_head_update_f32, _p(partial9), _p(bias2), _p(inv_depth), _p(disp_range),
                disp_range.numel(), h, w, _p(out_inv), _p(out_depth), _stream()), "effi_head_update_f32")
    return out_inv, out_depth


# =============================================================================================
# scope row n2: training kernels (csrc/train_ops.hip, warpcorr_dyn backward); thin wrappers, shapes checked here
# =============================================================================================
PW_ACT_BWD = {ACT_RELU: 0, ACT_SIGMOID: 1, ACT_TANH: 2}
PW_TANH, PW_RELU, PW_SIGMOID, PW_MUL, PW_MUL_BWD, PW_GRU, PW_GRU_BWD, PW_INV_TO_DEPTH, PW_INV_TO_DEPTH_BWD, PW_SCALE_CH = 3, 4, 5, 6, 7, 8, 9, 10, 11, 12


def conv_wgrad(a, b, dw, cb_off, kd, ks, stride=(1, 1)):
    """dw[ca][cb_off + cb][taps] += sum_o a[ca][o] * b[cb][o*stride + tap - pad]   (a on the small grid, b on the large one;
    [c,h,w] or [c,D,h,w]); see effi_conv_wgrad_f32."""
    _t(a, "wgrad a"), _t(b, "wgrad b"), _t(dw, "wgrad dw")
    if a.dim() == 3:
        a, b = a.unsqueeze(1), b.unsqueeze(1)
    ca, Da, ha, wa = a.shape
    cb, Db, hb, wb = b.shape
    sz, sxy = int(stride[0]), int(stride[1])
    if (Da, ha, wa) != ((Db - 1) // sz + 1 if kd == 3 else Db, (hb - 1) // sxy + 1, (wb - 1) // sxy + 1):
        raise ValueError(f"conv_wgrad: grids {tuple(a.shape)} / {tuple(b.shape)} do not match stride {stride}")
    if dw.shape[0] != ca or dw.numel() != ca * dw.shape[1] * kd * ks * ks:
        raise ValueError("conv_wgrad: dw shape")
    check(_lib.lib().effi_conv_wgrad_f32(_p(a), _p(b), ca, cb, dw.shape[1], cb_off, kd, ks, Da, ha, wa, Db, hb, wb, sz, sxy, _p(dw),
                                         _stream()), "effi_conv_wgrad_f32")
    return dw


def conv2d_k5s2_dgrad(grad_out, weight, hin, win):
    """Input gradient of a 5x5 / stride-2 / padding-2 convolution: grad_out [cout,ho,wo], weight [cout,cin,5,5] -> [cin,hin,win]."""
    _t(grad_out, "grad_out"), _t(weight, "weight")
    cout, cin = weight.shape[0], weight.shape[1]
    if tuple(grad_out.shape) != (cout, (hin - 1) // 2 + 1, (win - 1) // 2 + 1):
        raise ValueError("conv2d_k5s2_dgrad: grad_out shape does not match the input size")
    gx = torch.empty(cin, hin, win, device=grad_out.device, dtype=torch.float32)
    check(_lib.lib().effi_conv2d_k5s2_dgrad_f32(_p(grad_out), _p(weight), cin, cout, hin, win, _p(gx), _stream()), "effi_conv2d_k5s2_dgrad_f32")
    return gx


def conv2d_k5s2_dgrad_mfma(grad_out, weight, hin, win):
    """``conv2d_k5s2_dgrad`` on the matrix cores: one 3x3 convolution over ``grad_out`` whose 4 cin output channels are the four pixel
    parities of the input gradient (weights re-arranged on the device, effi_pack_conv2d_k5s2_dgrad_f32), then a pixel shuffle."""
    _t(grad_out, "grad_out"), _t(weight, "weight")
    cout, cin = weight.shape[0], weight.shape[1]
    ho, wo = (hin - 1) // 2 + 1, (win - 1) // 2 + 1
    if tuple(grad_out.shape) != (cout, ho, wo):
        raise ValueError("conv2d_k5s2_dgrad: grad_out shape does not match the input size")
    planes = torch.empty(4 * cin, ho, wo, device=grad_out.device, dtype=torch.float32)
    kg = (cout + 3) // 4
    for c0 in range(0, cin, 16):                       # at most 64 output channels per launch
        cn = min(16, cin - c0)
        nt = (4 * cn + 15) // 16
        if nt == 5:
            raise NotImplementedError("conv2d_k5s2_dgrad_mfma: input channels must come in groups of 4 / 8 / 12 / 16")
        buf = torch.empty(kg * 9 * nt * 64 + 16 * nt, device=grad_out.device, dtype=torch.float32)
        wp, bp = buf[:kg * 9 * nt * 64], buf[kg * 9 * nt * 64:]
        check(_lib.lib().effi_pack_conv2d_k5s2_dgrad_f32(_p(weight), cout, cin, c0, cn, _p(wp), _p(bp), _stream()),
              "effi_pack_conv2d_k5s2_dgrad_f32")
        conv2d([grad_out], wp, bp, 4 * cn, 3, out0=planes[4 * c0:4 * (c0 + cn)])
    gx = torch.nn.functional.pixel_shuffle(planes.unsqueeze(0), 2)[0]                # [cin, 2 ho, 2 wo]
    return gx if (2 * ho, 2 * wo) == (hin, win) else gx[:, :hin, :win].contiguous()


# elements per workgroup of the split per-channel reductions.  8192 left the chip short of workgroups (8 channels x 40 workgroups at
# 512x640: 1.25 per CU, 27 us for a 31-MB pass); 2048 = 8 elements per thread
# (option reduce_chunk; graph-replayed training step 28.6 -> 26.3 ms)

def _reduce_split(total, channels, k=1, device=None):
    """Workgroups per channel of the split per-channel reductions and their scratch ([C][nsplit][k] floats): ~2 K elements per
    workgroup, at most 256 per channel, none when the tensor is small.  The partial sums are added in a fixed order (second launch)."""
    nsplit = int(min(256, max(1, total // _PY_OPTS["reduce_chunk"])))
    if nsplit <= 1:
        return None, 1
    return torch.empty(channels * nsplit * k, device=device, dtype=torch.float32), nsplit


def channel_sum(g):
    """[B,C,...] -> [C] sums over batch and positions."""
    _t(g, "channel_sum input")
    B, Cc = g.shape[0], g.shape[1]
    n = g.numel() // (B * Cc)
    out = torch.empty(Cc, device=g.device, dtype=torch.float32)
    scratch, nsplit = _reduce_split(B * n, Cc, 1, g.device)
    check(_lib.lib().effi_channel_sum_f32(_p(g), B, Cc, n, _p(out), _p(scratch), nsplit, _stream()), "effi_channel_sum_f32")
    return out


def bn_moments(x):
    """[B,C,...] -> (mean [C], biased variance [C]) over batch and positions (two passes: sum, then centred squares)."""
    _t(x, "bn input")
    B, Cc = x.shape[0], x.shape[1]
    n = x.numel() // (B * Cc)
    mean = torch.empty(Cc, device=x.device, dtype=torch.float32)
    scratch, nsplit = _reduce_split(B * n, Cc, 1, x.device)
    check(_lib.lib().effi_bn_moment_f32(_p(x), B, Cc, n, None, 1, _p(mean), _p(scratch), nsplit, _stream()), "effi_bn_moment_f32")
    mean = mean / float(B * n)
    var = torch.empty(Cc, device=x.device, dtype=torch.float32)
    check(_lib.lib().effi_bn_moment_f32(_p(x), B, Cc, n, _p(mean), 2, _p(var), _p(scratch), nsplit, _stream()), "effi_bn_moment_f32")
    return mean, var / float(B * n)


def bn_train_fwd(x, gamma, beta, eps, momentum, running_mean=None, running_var=None, n_tracked=None, relu=False):
    """nn.BatchNorm (training mode) forward in one entry: -> (y, mean [C], invstd [C]); the running statistics (and the int64 counter
    ``n_tracked``) are updated in place when given."""
    _t(x, "bn input"), _t(gamma, "bn weight"), _t(beta, "bn bias")
    B, Cc = x.shape[0], x.shape[1]
    n = x.numel() // (B * Cc)
    nsplit = int(min(256, max(1, (B * n) // _PY_OPTS["reduce_chunk"])))
    buf = torch.empty(Cc * (2 * nsplit + 2), device=x.device, dtype=torch.float32)    # scratch (two passes) | mean | invstd in one allocation
    mean, invstd = buf[2 * Cc * nsplit:Cc * (2 * nsplit + 1)], buf[Cc * (2 * nsplit + 1):]
    y = torch.empty_like(x)
    if running_mean is not None:
        _t(running_mean, "running_mean"), _t(running_var, "running_var")
    if n_tracked is not None and (n_tracked.dtype != torch.int64 or n_tracked.device != x.device):
        raise ValueError("bn_train_fwd: num_batches_tracked must be an int64 tensor on the input's device")
    check(_lib.lib().effi_bn_train_fwd_f32(_p(x), B, Cc, n, _p(gamma), _p(beta), float(eps), float(momentum), _p(running_mean),
                                           _p(running_var), _p(n_tracked), int(relu), _p(y), _p(mean), _p(invstd), _p(buf), nsplit,
                                           _stream()), "effi_bn_train_fwd_f32")
    return y, mean, invstd


_PACK_CACHE = {}


def drop_pack_cache():
    """Forget every packed weight: needed after an update the version counters do not see (a replayed graph's optimizer step,
    writes through ``p.data``)."""
    _PACK_CACHE.clear()


def pack_conv2d_mfma_dev(weight, bias=None, dgrad=False):
    """``packing.pack_conv2d_mfma`` on the device in one launch (training: every layer, every step); ``dgrad``: the weights of the
    input-gradient convolution (``weight.flip(2, 3).transpose(0, 1)`` packed).  -> (wpack, bias_pack)."""
    _t(weight, "weight")
    cout, cin, ks, _ = weight.shape
    co, ci = (cin, cout) if dgrad else (cout, cin)
    if co == 1 and ks == 3:       # single-output-channel layers carry a second (vector) weight block: host form
        from . import packing
        w = weight.flip(2, 3).transpose(0, 1).contiguous() if dgrad else weight
        return packing.pack_conv2d_mfma(w, bias)
    # a layer is applied several times per step (five images through the feature pyramid, three GRU iterations per stage): its
    # packed weights are reused until the tensor changes (``_version`` counts in-place updates: the optimizer's) or dies
    # (entries made outside a graph capture are not valid inside one -- the replay must re-pack -- and vice versa)
    key = (id(weight), bool(dgrad), torch.cuda.is_current_stream_capturing())
    hit = _PACK_CACHE.get(key)
    if hit is not None and hit[0]() is weight and hit[1] == weight._version and hit[2] is bias and (bias is None or hit[3] == bias._version):
        return hit[4], hit[5]
    nt, kg = (co + 15) // 16, (ci + 3) // 4
    buf = torch.empty(kg * ks * ks * nt * 64 + 16 * nt, device=weight.device, dtype=torch.float32)
    wp, bp = buf[:kg * ks * ks * nt * 64].view(kg, ks * ks, nt, 64), buf[kg * ks * ks * nt * 64:]
    if bias is not None:
        _t(bias, "bias")
    check(_lib.lib().effi_pack_conv2d_mfma_f32(_p(weight), _p(bias), cout, cin, ks, int(dgrad), _p(wp), _p(bp), _stream()),
          "effi_pack_conv2d_mfma_f32")
    if len(_PACK_CACHE) > 4096:
        _PACK_CACHE.clear()
    _PACK_CACHE[key] = (weakref.ref(weight), weight._version, bias, None if bias is None else bias._version, wp, bp)
    return wp, bp


def bn_apply(x, mean, invstd, gamma, beta, relu):
    _t(x, "bn input")
    B, Cc = x.shape[0], x.shape[1]
    y = torch.empty_like(x)
    check(_lib.lib().effi_bn_apply_f32(_p(x), B, Cc, x.numel() // (B * Cc), _p(mean), _p(invstd), _p(gamma), _p(beta), int(relu), _p(y),
                                       _stream()), "effi_bn_apply_f32")
    return y


def bn_bwd(gy, y, x, mean, invstd, gamma, relu):
    """-> (gx, s1 = grad beta, s2 = grad gamma)."""
    _t(gy, "bn grad"), _t(y, "bn output"), _t(x, "bn input")
    B, Cc = x.shape[0], x.shape[1]
    s1 = torch.empty(Cc, device=x.device, dtype=torch.float32)
    s2 = torch.empty(Cc, device=x.device, dtype=torch.float32)
    gx = torch.empty_like(x)
    n = x.numel() // (B * Cc)
    scratch, nsplit = _reduce_split(B * n, Cc, 4, x.device)           # [C][nsplit][2] doubles
    check(_lib.lib().effi_bn_bwd_f32(_p(gy), _p(y), _p(x), B, Cc, n, _p(mean), _p(invstd), _p(gamma), int(relu),
                                     _p(s1), _p(s2), _p(gx), _p(scratch), nsplit, _stream()), "effi_bn_bwd_f32")
    return gx, s1, s2


def pointwise(op, a, b=None, c=None, d=None, s0=0.0, s1=0.0, inner=1, C=1, n_out=1):
    """One of the EFFI_PW_* element-wise ops over ``a.numel()`` elements -> 1..3 outputs shaped like ``a``."""
    for t_ in (a, b, c, d):
        if t_ is not None:
            _t(t_, "pointwise operand")
    outs = [torch.empty_like(a) for _ in range(n_out)]
    o = outs + [None] * (3 - n_out)
    check(_lib.lib().effi_pointwise_f32(op, _p(a), _p(b), _p(c), _p(d), float(s0), float(s1), a.numel(), inner, C, _p(o[0]), _p(o[1]),
                                        _p(o[2]), _stream()), "effi_pointwise_f32")
    return outs[0] if n_out == 1 else outs


def vol_lookup1d_bwd(gout, Dp, query, dmin, dmax, h, w):
    """Gradient of ``vol_lookup1d`` w.r.t. a planar [Dp,h,w] volume: gout [nq,h,w] -> [Dp,h,w]."""
    _t(gout, "grad"), _t(query, "query")
    nq, qh, qw = query.shape
    if (qh, qw) == (h, w):
        qys, qxs = qw, 1
    elif (qh // 2, qw // 2) == (h, w):
        qys, qxs = 2 * qw, 2
    else:
        raise ValueError("query resolution must equal the volume's or be twice it")
    dmin_t, rps = _range_ptr(dmin, h, w)
    dmax_t, _ = _range_ptr(dmax, h, w)
    gvol = torch.zeros(Dp, h, w, device=gout.device, dtype=torch.float32)
    check(_lib.lib().effi_vol_lookup1d_bwd_f32(_p(gvol), h * w, 1, Dp, _p(query), qh * qw, qys, qxs, nq, _p(dmin_t), _p(dmax_t), rps, h, w,
                                               _p(gout), _stream()), "effi_vol_lookup1d_bwd_f32")
    return gvol


def getcost_bwd(gcost, x, disp_range, interval, Dcur, Dreg, dmin, dmax, nq, h, w, input_is_depth=False):
    """Gradient of ``getcost`` w.r.t. planar cur [Dcur,h,w] and reg [Dreg,h,w] volumes: gcost [2nq,h,w] -> (gcur, greg)."""
    _t(gcost, "grad"), _t(x, "inv_depth"), _t(interval, "interval")
    dmin_t, gps = _range_ptr(dmin, h, w)
    dmax_t, _ = _range_ptr(dmax, h, w)
    gcur = torch.zeros(Dcur, h, w, device=x.device, dtype=torch.float32)
    greg = torch.zeros(Dreg, h, w, device=x.device, dtype=torch.float32)
    n_range = 0 if disp_range is None else disp_range.numel()
    check(_lib.lib().effi_getcost_bwd_f32(_p(x), _p(disp_range), n_range, int(input_is_depth), _p(interval), _p(gcur), h * w, 1, Dcur,
                                          _p(greg), h * w, 1, Dreg, _p(dmin_t), _p(dmax_t), gps, nq, h, w, _p(gcost), _stream()),
          "effi_getcost_bwd_f32")
    return gcur, greg


def softargmin_bwd(logits, hyp, gdepth):
    D, h, w = logits.shape
    _t(logits, "logits"), _t(hyp, "hypotheses", contiguous=False), _t(gdepth, "grad")
    hyp, dds, dps = _depth_strides(hyp, D, h, w)
    gl = torch.empty_like(logits)
    check(_lib.lib().effi_softargmin_bwd_f32(_p(logits), _p(hyp), dds, dps, D, h * w, _p(gdepth), _p(gl), _stream()), "effi_softargmin_bwd_f32")
    return gl


def view_aggregate_bwd(sim_views, weights, gout):
    S, D, h, w = sim_views.shape
    _t(sim_views, "sim_views"), _t(weights, "weights"), _t(gout, "grad")
    gsim, gw = torch.empty_like(sim_views), torch.empty_like(weights)
    check(_lib.lib().effi_view_aggregate_bwd_f32(_p(sim_views), _p(weights), S, D, h * w, _p(gout), _p(gsim), _p(gw), _stream()),
          "effi_view_aggregate_bwd_f32")
    return gsim, gw


def convex_upsample2x_bwd(inv_depth, mask, gup):
    _t(inv_depth, "inv_depth"), _t(mask, "mask"), _t(gup, "grad")
    h, w = inv_depth.shape[-2:]
    gmask = torch.empty_like(mask)
    ginv = torch.empty_like(inv_depth)
    scratch = torch.empty(9, h, w, device=inv_depth.device, dtype=torch.float32)       # per-pixel contributions, gathered in a fixed order
    check(_lib.lib().effi_convex_upsample2x_bwd_f32(_p(inv_depth), _p(mask), h, w, _p(gup), _p(gmask), _p(ginv), _p(scratch), _stream()),
          "effi_convex_upsample2x_bwd_f32")
    return gmask, ginv


def warpcorr_dyn_bwd(ref_nhwc, srcs_nhwc, rt, cur_depth, interval, view_w, D, sim, grad_sim):
    """Backward of ``warpcorr_dyn``: -> (grad_ref [h,w,C], [grad_src_v [h,w,C]], grad_view_w [S,vh,vw])."""
    h, w, Cc = ref_nhwc.shape
    S = len(srcs_nhwc)
    for t_ in (ref_nhwc, rt, cur_depth, interval, view_w, sim, grad_sim):
        _t(t_, "warpcorr_dyn_bwd operand")
    vh = view_w.shape[1]
    shift = 0
    while (vh << shift) < h:
        shift += 1
    g_ref = torch.empty_like(ref_nhwc)
    g_src = [torch.zeros_like(s_) for s_ in srcs_nhwc]
    g_vw = torch.zeros(view_w.shape, device=view_w.device, dtype=torch.float64)      # 64-bit atomics: the sum cancels (see the kernel)
    check(_lib.lib().effi_warpcorr_dyn_bwd_f32(_p(ref_nhwc), _ptr_array(srcs_nhwc), S, _p(rt), _p(cur_depth), _p(interval), _p(view_w), shift,
                                               Cc, h, w, D, _p(sim), _p(grad_sim), _p(g_ref), _ptr_array(g_src), _p(g_vw), _stream()),
          "effi_warpcorr_dyn_bwd_f32")
    return g_ref, g_src, g_vw.float()
