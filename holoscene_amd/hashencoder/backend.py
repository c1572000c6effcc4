"""ctypes binding of libholoscene_hip.so -- the MI355X stand-in for the reference's
``hashencoder/backend.py`` (which JIT-builds the pybind11 module ``_hash_encoder``,
backend.py:12-24).

``_backend`` exposes the same three callables with the same positional signature as the
reference module (hashencoder/src/bindings.cpp:5-9), taking torch tensors, plus the
strided variants the fused Python layer uses.  There is NO CPU or PyTorch fallback: a
missing library, a CPU tensor or a non-fp32 tensor raises ``RuntimeError`` (the reference
raises the same type from TORCH_CHECK, hashencoder.cu:16-19).
"""
import ctypes
import os

import torch

_CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "csrc")
LIB_PATH = os.environ.get("HOLOSCENE_LIB") or os.path.join(_CSRC, "libholoscene_hip.so")      # (HOLOSCENE_LIB: a variant build, csrc/build.py --variant; kernel A/B runs)

_ERRORS = {-1: "unsupported D/C/size combination (GridEncoding: D must be 2 or 3, C must be 1, 2, 4, or 8)",
           -2: "HIP kernel launch failed", -3: "required pointer was NULL"}


class hsGate(ctypes.Structure):
    """Device-side launch gate (include/holoscene_hip.h): run only if *a > *b."""
    _fields_ = [("a", ctypes.c_void_p), ("b", ctypes.c_void_p)]


class hsHashLayout(ctypes.Structure):
    _fields_ = [("level_stride", ctypes.c_int64), ("point_stride", ctypes.c_int64), ("dydx_level_stride", ctypes.c_int64),
                ("dydx_point_stride", ctypes.c_int64), ("schedule", ctypes.c_int32), ("gate", hsGate), ("scatter_ws", ctypes.c_void_p),
                ("scatter_cap", ctypes.c_uint32), ("grid_id", ctypes.c_void_p), ("grid_stride", ctypes.c_int64), ("ws_clean", ctypes.c_int32),
                ("out_bf16", ctypes.c_int32), ("r1_ux", ctypes.c_void_p), ("r1_g", ctypes.c_void_p), ("r1_n", ctypes.c_uint32), ("r1_scale", ctypes.c_float),
                ("step", ctypes.c_void_p)]


class hsTableStep(ctypes.Structure):
    _fields_ = [("p", ctypes.c_void_p), ("m", ctypes.c_void_p), ("v", ctypes.c_void_p), ("state", ctypes.c_void_p), ("beta1", ctypes.c_float),
                ("beta2", ctypes.c_float), ("eps", ctypes.c_float), ("grad_scale", ctypes.c_float), ("group", ctypes.c_int32), ("prior", ctypes.c_int32)]


class hsPackJob(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("ld", ctypes.c_int32), ("row0", ctypes.c_int32), ("col0", ctypes.c_int32),
                ("rows", ctypes.c_int32), ("cols", ctypes.c_int32), ("dst_rows", ctypes.c_int32), ("dst_cols", ctypes.c_int32),
                ("transpose", ctypes.c_int32), ("scale", ctypes.c_float)]


class hsTrunkSplit(ctypes.Structure):
    _fields_ = [("n_main", ctypes.c_int64), ("sdf_raw", ctypes.c_void_p), ("sdf", ctypes.c_void_p), ("idx", ctypes.c_void_p), ("grad", ctypes.c_void_p),
                ("y_eik", ctypes.c_void_p), ("min_eik", ctypes.c_void_p), ("grad_theta", ctypes.c_void_p)]


class hsWgradJob(ctypes.Structure):
    _fields_ = [("A", ctypes.c_void_p), ("B", ctypes.c_void_p), ("part", ctypes.c_void_p), ("M", ctypes.c_int64), ("NA", ctypes.c_int32), ("MB", ctypes.c_int32)]


class hsSumJob(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("n", ctypes.c_int64), ("slices", ctypes.c_int32), ("src_f32", ctypes.c_int32)]


class hsCopyJob(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("n", ctypes.c_int64)]


class hsWnJob(ctypes.Structure):
    _fields_ = [("v", ctypes.c_void_p), ("g", ctypes.c_void_p), ("gW", ctypes.c_void_p), ("W", ctypes.c_void_p), ("gv", ctypes.c_void_p),
                ("gg", ctypes.c_void_p), ("rows", ctypes.c_int32), ("cols", ctypes.c_int32)]


class hsWgradPairJob(ctypes.Structure):
    _fields_ = [("A0", ctypes.c_void_p), ("B0", ctypes.c_void_p), ("A1", ctypes.c_void_p), ("B1", ctypes.c_void_p), ("part", ctypes.c_void_p),
                ("M", ctypes.c_int64), ("rows", ctypes.c_int64), ("kind", ctypes.c_int32), ("slices", ctypes.c_int32), ("ones", ctypes.c_int32),
                ("reserved", ctypes.c_int32), ("colsum", ctypes.c_void_p)]


class hsAsmTerm(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p), ("col_map", ctypes.c_void_p), ("ld", ctypes.c_int64), ("red_stride", ctypes.c_int64), ("col0", ctypes.c_int32),
                ("red", ctypes.c_int32), ("src_bf16", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class hsAsmJob(ctypes.Structure):
    _fields_ = [("dst", ctypes.c_void_p), ("dst_ld", ctypes.c_int64), ("rows", ctypes.c_int32), ("cols", ctypes.c_int32), ("n_terms", ctypes.c_int32),
                ("reserved", ctypes.c_int32), ("term", hsAsmTerm * 3)]


class hsGatherJob(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("idx", ctypes.c_void_p), ("n", ctypes.c_int64), ("row_bytes", ctypes.c_int32)]


GATHER_MAX_JOBS = 12        # HS_GATHER_MAX_JOBS


class hsFrameDesc(ctypes.Structure):
    _fields_ = [("class_ptr", ctypes.c_void_p), ("class_pix", ctypes.c_void_p), ("out_off", ctypes.c_void_p), ("n_cls", ctypes.c_int32),
                ("per_class", ctypes.c_int32), ("n_bg", ctypes.c_int32), ("reserved", ctypes.c_int32), ("src", ctypes.c_void_p * GATHER_MAX_JOBS)]


class hsDrawSched(ctypes.Structure):
    _fields_ = [("frames", ctypes.c_void_p), ("sched", ctypes.c_void_p), ("cursor", ctypes.c_void_p), ("seed", ctypes.c_uint64),
                ("counter_base", ctypes.c_uint64), ("n_sched", ctypes.c_int32), ("n_frames", ctypes.c_int32)]


ABI_VERSION = 9

# Small zero-initialised accumulators (bias-gradient sums the backward kernels add to by atomics): slices of a pool the optimiser zeroes
# together with the flat gradient buffer -- one memset per iteration instead of one ~5 us fill launch per accumulator.  The sequence of
# requests inside an iteration is the same every time, so a captured graph sees the same addresses as its warm-up passes.
_ZERO_POOL = {"buf": None, "pos": 0}
_SCATTER_WS = {}


def set_zero_pool(buf):
    """buf: fp32 device tensor that the caller has just zeroed (training/flat.py: FlatAdam.zero_grad), or None to switch the pool off."""
    _ZERO_POOL["buf"], _ZERO_POOL["pos"] = buf, 0


def _pool_device(device):
    d = torch.device(device)
    return torch.device("cuda", torch.cuda.current_device()) if d.type == "cuda" and d.index is None else d


def zero_pool_armed(device, n=1):
    """Whether zeros_small(n, device) would be served from the pool (i.e. without a fill launch)."""
    buf = _ZERO_POOL["buf"]
    return buf is not None and buf.device == _pool_device(device) and _ZERO_POOL["pos"] + n <= buf.numel()




def zeros_small(n, device):
    """n zero floats: from the pool when one is armed on that device and has room, else a fresh torch.zeros."""
    buf, pos = _ZERO_POOL["buf"], _ZERO_POOL["pos"]
    if buf is not None and buf.device == _pool_device(device) and pos + n <= buf.numel():
        _ZERO_POOL["pos"] = pos + (n + 3) // 4 * 4
        return buf[pos:pos + n]
    return torch.zeros(n, device=device)



def _gate(gate):
    """gate: None or (a, b) one-element float32 device tensors -> hsGate (kept alive by the caller's references)."""
    if gate is None:
        return hsGate(None, None)
    a, b = gate
    return hsGate(_dev(a, "gate.a").value, _dev(b, "gate.b").value)


_lib = None
_DTYPES = {torch.float32: 0, torch.bfloat16: 1}  # HS_F32 / HS_BF16


def load_library():
    """dlopen the HIP library (built by holoscene_amd.csrc.build / __graft_entry__.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python -m holoscene_amd.csrc.build` (hipcc, gfx950). "
                               "There is no CPU fallback for the hash encoder.")
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.hs_target_arch.restype = ctypes.c_char_p
        for name in dir_symbols():
            if name != "hs_target_arch":
                getattr(_lib, name).restype = ctypes.c_int
        if _lib.hs_abi_version() != ABI_VERSION:
            raise RuntimeError(f"{LIB_PATH}: ABI version {_lib.hs_abi_version()}, this package needs {ABI_VERSION} -- rebuild "
                               "(python -m holoscene_amd.csrc.build)")
    return _lib


def dir_symbols():
    """Every symbol include/holoscene_hip.h declares (kept in sync by tests/test_abi.py)."""
    return ["hs_abi_version", "hs_target_arch", "hs_hash_encode_forward", "hs_hash_encode_backward", "hs_hash_encode_second_backward",
            "hs_hash_fwd", "hs_hash_bwd", "hs_hash_bwd2", "hs_hash_bwd_jac", "hs_hash_encode_forward_dt", "hs_hash_encode_backward_dt", "hs_hash_encode_second_backward_dt", "hs_hash_scatter_ws_bytes", "hs_sampler_update", "hs_sampler_draw", "hs_sampler_final", "hs_sampler_step", "hs_sampler_pick", "hs_sampler_draw_step", "hs_sampler_draw_steps", "hs_sampler_tail", "hs_sampler_update_draw", "hs_softplus_tangent_fwd", "hs_softplus_tangent_bwd", "hs_adam_tick", "hs_adam_flat", "hs_adam_flat_shard", "hs_copy_many", "hs_composite_fwd", "hs_composite_bwd", "hs_sdf_mlp_fwd", "hs_sdf_mlp2_pack_bytes", "hs_sdf_mlp2_pack", "hs_sdf_mlp2_fwd", "hs_sdf_mlp2_fwd_wide", "hs_sdf_sweep_fwd", "hs_sdf_mlp32_pack_bytes", "hs_sdf_mlp32_pack", "hs_sdf_mlp32_fwd", "hs_trunk_mlp2_input_column", "hs_trunk_mlp2_fwd", "hs_trunk_mlp2_fwd_wide", "hs_trunk_mlp_fwd", "hs_trunk_mlp_bwd", "hs_trunk_bwd_parts", "hs_trunk_split_fwd", "hs_trunk_split_bwd", "hs_softplus_tangent_bwd_h", "hs_trunk_input_fwd", "hs_trunk_input_bwd", "hs_render_input_fwd", "hs_render_input_bwd", "hs_loss_rays", "hs_loss_eikonal", "hs_loss_stage1", "hs_bg_smooth_loss", "hs_ray_setup", "hs_ray_points", "hs_render_points", "hs_appearance_mask_words", "hs_appearance_fwd", "hs_appearance_bwd", "hs_pack_bf16", "hs_sum_slices", "hs_weight_norm", "hs_gather_rows", "hs_wgrad_rows", "hs_draw_pixels", "hs_draw_gather", "hs_draw_gather_sched", "hs_hash_bwd_draw", "hs_hash_bwd_jac_sums", "hs_iter_prologue", "hs_iter_prologue_draw", "hs_iter_epilogue", "hs_pack_iteration", "hs_trunk_rr_gy", "hs_trunk_rr_gy_split", "hs_trunk_rr_pack_bytes", "hs_trunk_rr_pack", "hs_trunk_rr_fwd_value", "hs_trunk_rr_fwd_wide", "hs_trunk_rr_bwd_value_wide",
            "hs_trunk_rr_fwd_grad", "hs_trunk_rr_fwd", "hs_trunk_rr_bwd_grad", "hs_trunk_rr_bwd_value", "hs_wgrad_pairs", "hs_assemble", "hs_abs_shift", "hs_trunk_pack_all", "hs_appearance2_pack_bytes", "hs_appearance2_enc_column", "hs_appearance2_pack",
            "hs_appearance2_fwd", "hs_appearance2_pack_t_bytes", "hs_appearance2_bwd", "hs_gemm_split_nt", "hs_gemm_split_tn"]


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what}: {_ERRORS.get(rc, rc)}")


def _dev(t, name, dtype=torch.float32):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous tensor")
    if t.dtype != dtype:
        raise RuntimeError(f"{name} must be a {dtype} tensor")
    return ctypes.c_void_p(t.data_ptr())


def _dev_at(t, name, elem_offset, dtype=torch.float32):
    """_dev(t) advanced by elem_offset elements (a sub-range of a level-major buffer: the kernels take a level stride `ld`)."""
    p = _dev(t, name, dtype)
    return None if p is None else ctypes.c_void_p(p.value + int(elem_offset) * t.element_size())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


SCHEDULE = int(os.environ.get("HOLOSCENE_HASH_SCHEDULE", "1"))
# Scatter the hashed levels through per-bin record lists + an LDS reduction instead of global atomics (csrc/hash_encode.hip)
SCATTER_BINS = True

def accumulates_into_grad(table):
    """True for a hash table whose gradient lives in flat gradient storage (training/flat.py marks the parameter with
    `_hs_flat_owner` and keeps its `.grad` view attached): the scatter kernels then accumulate straight into that buffer and
    the autograd Functions return no table gradient -- this removes the per-call 48.8 MB zero-fill + AccumulateGrad add of
    the reference (hashgrid.py:75-76).  Per table, not per process: any other model in the same process keeps the plain
    torch.autograd.grad(..., embeddings) semantics."""
    return table is not None and getattr(table, "_hs_flat_owner", False) and table.grad is not None


class ScatterWatch:
    """Tells a listener when a flat-owned table's gradient is FINAL inside a backward pass (data parallelism: the table's segment
    can be exchanged from that moment on, training/trainer.py).  Every autograd Function that will scatter into the table in
    place calls `expect_scatter` in its forward (only when the table requires a gradient there) and `scatter_done` after the
    scatter; the callback runs -- on the autograd thread, with the scatter's stream current -- when the count returns to zero.
    A producer that takes another route to the gradient (double backward) never reports done, so the callback simply does
    not fire and the listener exchanges the segment at the end of the pass.  Every Function that scatters into a table in
    place reports (model/network.py: _hash_encode_jac, _trunk_input, _fused_trunk, _fused_trunk_render, _fused_appearance;
    hashgrid.py: _hash_encode), so a watched table never has an uncounted in-place producer."""

    def __init__(self):
        self.pending, self.callback = 0, None

    def arm(self, callback):
        self.pending, self.callback = 0, callback


def expect_scatter(table):
    w = getattr(table, "_hs_scatter_watch", None) if table is not None else None
    if w is not None and accumulates_into_grad(table):
        w.pending += 1


def scatter_done(table):
    w = getattr(table, "_hs_scatter_watch", None) if table is not None else None
    if w is not None and w.pending > 0:
        w.pending -= 1
        if w.pending == 0 and w.callback is not None:
            cb, w.callback = w.callback, None
            cb()


# Reduce-and-step (hsTableStep, include/holoscene_hip.h): data_ptr of a flat-owned table's gradient view -> [hsTableStep, scatters served,
# scatters expected].  training/flat.py registers the tables for the span of ONE backward pass (FlatAdam.table_steps: the entries exist inside
# that context only) with the number of gradient producers the trainer counted for the variant; the scatter wrappers below let the earlier
# producers accumulate into the gradient table the plain way and attach the step (hsTableStep.prior = "add what is there") to the LAST one.
# One more scatter than expected would miss the step, and a last producer that cannot carry it (double backward, the reference-compatible
# trio) would leave the table unstepped: both raise.
TABLE_STEPS = {}
SCATTER_COUNTS = None       # a dict while a trainer counts the gradient producers per table (training/trainer.py)


def _table_step(grad_embeddings, can_step=True):
    if grad_embeddings is None:
        return None
    ptr = grad_embeddings.data_ptr()
    if SCATTER_COUNTS is not None:
        SCATTER_COUNTS[ptr] = SCATTER_COUNTS.get(ptr, 0) + 1
    ent = TABLE_STEPS.get(ptr)
    if ent is None:
        return None
    n = ent[1] + 1
    if n < ent[2]:
        ent[1] = n
        return None             # an earlier producer: plain accumulation into the (all-zero on entry) gradient table
    if n > ent[2] or not can_step:
        raise RuntimeError("a hash table registered for reduce-and-step received more gradient producers in this backward pass than were counted "
                           "for it, or its last producer cannot carry the step: its optimiser step must go through FlatAdam.step instead")
    ent[1] = n
    return ent[0]


def point_major_layout(C, L, D):
    """features [B, L*C]; dy_dx [L, B, D*C] (level stride filled in per call)."""
    return dict(level_stride=C, point_stride=L * C, dydx_point_stride=D * C)


class _HipBackend:
    # ---- reference-compatible trio (bindings.cpp:5-9)
    @staticmethod
    def hash_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, calc_grad_inputs, dy_dx):
        lib = load_library()
        _check(lib.hs_hash_encode_forward(_dev(inputs, "inputs"), _dev(embeddings, "embeddings"), _dev(offsets, "offsets", torch.int32),
                                          _dev(outputs, "outputs"), B, D, C, L, ctypes.c_float(S), H, int(bool(calc_grad_inputs)),
                                          _dev(dy_dx, "dy_dx"), _stream()), "hash_encode_forward")

    @staticmethod
    def hash_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, calc_grad_inputs, dy_dx, grad_inputs):
        lib = load_library()
        _table_step(grad_embeddings, can_step=False)
        _check(lib.hs_hash_encode_backward(_dev(grad, "grad"), _dev(inputs, "inputs"), _dev(embeddings, "embeddings"),
                                           _dev(offsets, "offsets", torch.int32), _dev(grad_embeddings, "grad_embeddings"), B, D, C, L,
                                           ctypes.c_float(S), H, int(bool(calc_grad_inputs)), _dev(dy_dx, "dy_dx"),
                                           _dev(grad_inputs, "grad_inputs"), _stream()), "hash_encode_backward")

    @staticmethod
    def hash_encode_second_backward(grad, inputs, embeddings, offsets, B, D, C, L, S, H, calc_grad_inputs, dy_dx, grad_grad_inputs,
                                    grad_grad, grad2_embeddings):
        lib = load_library()
        _table_step(grad2_embeddings, can_step=False)
        _check(lib.hs_hash_encode_second_backward(_dev(grad, "grad"), _dev(inputs, "inputs"), _dev(embeddings, "embeddings"),
                                                  _dev(offsets, "offsets", torch.int32), B, D, C, L, ctypes.c_float(S), H,
                                                  int(bool(calc_grad_inputs)), _dev(dy_dx, "dy_dx"),
                                                  _dev(grad_grad_inputs, "grad_grad_inputs"), _dev(grad_grad, "grad_grad"),
                                                  _dev(grad2_embeddings, "grad2_embeddings"), _stream()), "hash_encode_second_backward")

    # ---- strided / selective variants (include/holoscene_hip.h section 2); point-major features, level-major dy_dx
    @staticmethod
    def _layout(B, D, C, L, gate=None, ws=None, level_major=False, grids=None):
        """grids: None, or (grid_id int32 [B], entries per grid): a batched-over-grids launch (hsHashLayout::grid_id)."""
        lay = hsHashLayout(C, L * C, B * D * C, D * C, SCHEDULE, _gate(gate), None, 0, None, 0, 0, 0, None, None, 0, 0.0, None)
        if grids is not None:
            if ws is not None:
                raise ValueError("the binned scatter holds the records of ONE table: no scatter work space with grids=")
            gid, stride = grids
            if gid.numel() != B:
                raise ValueError("grid_id: one entry per point")
            lay.grid_id, lay.grid_stride = _dev(gid, "grid_id", torch.int32).value, int(stride)
        if level_major:      # features [L, B, C]: consecutive lanes (points) write consecutive 4C-byte entries
            lay.level_stride, lay.point_stride = B * C, C
        if ws is not None:
            buf, cap = ws[:2]
            lay.scatter_ws, lay.scatter_cap = _dev(buf, "scatter_ws", torch.uint8).value, cap
            lay.ws_clean = int(len(ws) > 2 and bool(ws[2]))
        return lay

    @staticmethod
    def scatter_workspace(B, D, C, L, device):
        """(uint8 buffer, per-bin capacity) for the binned scatter of hs_hash_bwd / hs_hash_bwd_jac, or None when disabled."""
        if not SCATTER_BINS:
            return None
        lib = load_library()
        lib.hs_hash_scatter_ws_bytes.restype = ctypes.c_int64
        cap = ctypes.c_uint32(0)
        n = lib.hs_hash_scatter_ws_bytes(B, D, C, L, ctypes.byref(cap))
        if n < 0:
            raise RuntimeError(f"hs_hash_scatter_ws_bytes: {n}")
        # ONE persistent work space per (D, C, L, device), zero-filled once and sized for the largest batch seen so far, rounded up to a power
        # of two (a smaller batch runs through the same buffer with the larger per-bin capacity: the kernels take the capacity from the
        # layout, not from B) -- the eager path's batch size varies per frame, and one ~2.3 KB x B buffer per distinct B would pin memory
        # until the process dies.  The reduce kernel returns the bin counters to zero, so a scatter through a CLEAN work space needs no
        # clearing launch (third element -> hsHashLayout::ws_clean); that only holds while every scatter through it is ordered on one
        # stream, so a call from another stream than the last one gets the clearing launch (and takes the buffer over).
        # Created eagerly (the warm-up passes of a captured iteration), never inside a capture.
        key = (int(D), int(C), int(L), str(torch.device(device)))
        stream = int(torch.cuda.current_stream(device).cuda_stream)
        hit = _SCATTER_WS.get(key)
        if hit is None or hit["B"] < B:
            if torch.cuda.is_current_stream_capturing():
                return torch.empty(n, device=device, dtype=torch.uint8), int(cap.value)
            Bcap = 1 << max(int(B) - 1, 1).bit_length()
            n2 = lib.hs_hash_scatter_ws_bytes(Bcap, D, C, L, ctypes.byref(cap))
            if n2 < 0:
                raise RuntimeError(f"hs_hash_scatter_ws_bytes: {n2}")
            _SCATTER_WS.pop(key, None)          # (the smaller buffer goes back to the allocator)
            hit = _SCATTER_WS[key] = {"B": Bcap, "buf": torch.zeros(n2, device=device, dtype=torch.uint8), "cap": int(cap.value), "stream": stream}
        # (stream None: the owner declared the device idle since the last use -- scatter_workspaces_idle(), the trainer after the
        #  synchronize that ends its warm-up passes -- so whoever comes next, the capture stream included, finds the counters at zero)
        clean = hit["stream"] is None or hit["stream"] == stream
        hit["stream"] = stream
        return hit["buf"], hit["cap"], clean

    @staticmethod
    def scatter_workspaces_idle():
        """Declare that every scatter issued so far has completed (the caller has synchronised the device): the next scatter through a
        cached work space may come from any stream without the clearing launch."""
        for hit in _SCATTER_WS.values():
            hit["stream"] = None

    @classmethod
    def fwd(cls, inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx, gate=None, level_major=False, grids=None, out_bf16=False):
        """out_bf16: outputs is an int32 [L, B] tensor that receives one word per (level, point) = both channels as bf16 (C == 2, no dy_dx):
        the form hs_sdf_mlp2_fwd(feat_level_major=2) consumes -- half the bytes of the fp32 features, written and read."""
        lib = load_library()
        lay = cls._layout(B, D, C, L, gate, level_major=level_major, grids=grids)
        if out_bf16:
            if C != 2 or dy_dx is not None or outputs.dtype != torch.int32 or tuple(outputs.shape) != (L, B):
                raise RuntimeError("hash fwd out_bf16: int32 [L, B] outputs, C == 2, no dy_dx")
            lay.out_bf16, lay.level_stride, lay.point_stride = 1, B, 1
        _check(lib.hs_hash_fwd(_dev(inputs, "inputs"), _dev(embeddings, "embeddings"), _dev(offsets, "offsets", torch.int32),
                               _dev(outputs, "outputs", torch.int32 if out_bf16 else torch.float32), B, D, C, L, ctypes.c_float(S), H, _dev(dy_dx, "dy_dx"),
                               ctypes.byref(lay), _stream()), "hs_hash_fwd")

    @classmethod
    def bwd(cls, grad, inputs, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx, grad_inputs, ws=None, level_major=False, grids=None, rider=None, sums=None):
        """Workgroups riding in front of this scatter's (hs_hash_bwd_draw) -- rider: None or ScheduledDraw.args() = (draw_sched_plan(), n_uniform,
        total_pixels, n_out): the NEXT iteration's batch; sums: None or what assemble(jobs, defer=True) returned beside the results: that launch."""
        lib = load_library()
        lay = cls._layout(B, D, C, L, ws=ws, level_major=level_major, grids=grids)
        step = _table_step(grad_embeddings)
        if step is not None:
            lay.step = ctypes.addressof(step)
        head = (_dev(grad, "grad"), _dev(inputs, "inputs"), _dev(offsets, "offsets", torch.int32),
                _dev(grad_embeddings, "grad_embeddings"), B, D, C, L, ctypes.c_float(S), H, _dev(dy_dx, "dy_dx"),
                _dev(grad_inputs, "grad_inputs"), ctypes.byref(lay))
        if rider is None and not sums:
            _check(lib.hs_hash_bwd(*head, _stream()), "hs_hash_bwd")
        else:
            sarr, n_sums = cls._asm_array(sums)
            if rider is None:
                dargs = (None, 0, 0, 0, None, None, 0)
            else:
                (dst, darr, n_jobs, keep), n_uniform, total_pixels, n_out = rider
                dargs = (ctypes.byref(dst), int(n_uniform), int(total_pixels), int(n_out), _dev(keep[3], "out", torch.int64), darr, n_jobs)
            _check(lib.hs_hash_bwd_draw(*head, *dargs, sarr, n_sums, _stream()), "hs_hash_bwd_draw")
            if sums:
                sums.clear()

    @classmethod
    def bwd2(cls, grad, inputs, offsets, B, D, C, L, S, H, dy_dx, grad_grad_inputs, grad_grad, grad2_embeddings, grids=None):
        lib = load_library()
        lay = cls._layout(B, D, C, L, grids=grids)
        _table_step(grad2_embeddings, can_step=False)
        _check(lib.hs_hash_bwd2(_dev(grad, "grad"), _dev(inputs, "inputs"), _dev(offsets, "offsets", torch.int32), B, D, C, L,
                                ctypes.c_float(S), H, _dev(dy_dx, "dy_dx"), _dev(grad_grad_inputs, "grad_grad_inputs"),
                                _dev(grad_grad, "grad_grad"), _dev(grad2_embeddings, "grad2_embeddings"), ctypes.byref(lay),
                                _stream()), "hs_hash_bwd2")

    # ---- the reference's three entry points in its other scalar types (include/holoscene_hip.h: *_dt; csrc/hash_encode_dt.hip).  Reference layouts:
    # outputs / grad [L, B, C], dy_dx [B, L * D * C]; every tensor of the inputs' dtype (torch.float64 / float16; float32 forwards to the functions above)
    DTYPES = {torch.float32: 0, torch.float64: 1, torch.float16: 2}

    @classmethod
    def encode_forward_dt(cls, inputs, embeddings, offsets, outputs, B, D, C, L, S, H, dy_dx=None):
        lib, dt = load_library(), inputs.dtype
        _check(lib.hs_hash_encode_forward_dt(cls.DTYPES[dt], _dev(inputs, "inputs", dt), _dev(embeddings, "embeddings", dt), _dev(offsets, "offsets", torch.int32),
                                             _dev(outputs, "outputs", dt), B, D, C, L, ctypes.c_float(S), H, int(dy_dx is not None), _dev(dy_dx, "dy_dx", dt),
                                             _stream()), "hs_hash_encode_forward_dt")

    @classmethod
    def encode_backward_dt(cls, grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, dy_dx=None, grad_inputs=None):
        lib, dt = load_library(), inputs.dtype
        _check(lib.hs_hash_encode_backward_dt(cls.DTYPES[dt], _dev(grad, "grad", dt), _dev(inputs, "inputs", dt), _dev(embeddings, "embeddings", dt),
                                              _dev(offsets, "offsets", torch.int32), _dev(grad_embeddings, "grad_embeddings", dt), B, D, C, L, ctypes.c_float(S), H,
                                              int(grad_inputs is not None), _dev(dy_dx, "dy_dx", dt), _dev(grad_inputs, "grad_inputs", dt), _stream()),
               "hs_hash_encode_backward_dt")

    @classmethod
    def encode_second_backward_dt(cls, grad, inputs, embeddings, offsets, B, D, C, L, S, H, dy_dx, grad_grad_inputs, grad_grad, grad2_embeddings):
        lib, dt = load_library(), inputs.dtype
        _check(lib.hs_hash_encode_second_backward_dt(cls.DTYPES[dt], _dev(grad, "grad", dt), _dev(inputs, "inputs", dt), _dev(embeddings, "embeddings", dt),
                                                     _dev(offsets, "offsets", torch.int32), B, D, C, L, ctypes.c_float(S), H, 1, _dev(dy_dx, "dy_dx", dt),
                                                     _dev(grad_grad_inputs, "grad_grad_inputs", dt), _dev(grad_grad, "grad_grad", dt),
                                                     _dev(grad2_embeddings, "grad2_embeddings", dt), _stream()), "hs_hash_encode_second_backward_dt")

    @classmethod
    def bwd_jac(cls, g_feat, g_dydx, inputs, offsets, grad_embeddings, B, D, C, L, S, H, ws=None, level_major=False, grids=None, rank1=None, sums=None):
        """rank1: None, or (ux fp32 [n, L*C], g fp32 [n, D], scale): the dy_dx cotangent of the first n points is (scale * ux[b, l*C + c]) * g[b, d]
        and is formed inside the kernel; those rows of g_dydx are not read (g_dydx may be None when n == B).  sums: None or what
        assemble(jobs, defer=True) returned beside the results: that launch rides in front of this scatter's workgroups (hs_hash_bwd_jac_sums)."""
        lib = load_library()
        lay = cls._layout(B, D, C, L, ws=ws, level_major=level_major, grids=grids)
        if rank1 is not None:
            ux, g1, scale = rank1
            if ux.shape[0] != g1.shape[0] or ux.shape[0] > B or ux.shape[1] != L * C or g1.shape[1] != D:
                raise RuntimeError("bwd_jac rank1: ux [n, L*C], g [n, D], n <= B")
            lay.r1_ux, lay.r1_g, lay.r1_n, lay.r1_scale = _dev(ux, "rank1 ux").value, _dev(g1, "rank1 g").value, int(ux.shape[0]), float(scale)
        step = _table_step(grad_embeddings)
        if step is not None:
            lay.step = ctypes.addressof(step)
        sarr, n_sums = cls._asm_array(sums)
        _check(lib.hs_hash_bwd_jac_sums(_dev(g_feat, "g_feat"), _dev(g_dydx, "g_dydx"), _dev(inputs, "inputs"),
                                        _dev(offsets, "offsets", torch.int32), _dev(grad_embeddings, "grad_embeddings"), B, D, C, L,
                                        ctypes.c_float(S), H, ctypes.byref(lay), sarr, n_sums, _stream()), "hs_hash_bwd_jac_sums")
        if sums:
            sums.clear()

    @staticmethod
    def _asm_array(pending):
        """assemble(defer=True)'s pending jobs as hs_assemble's argument pair (None, 0 when there is nothing)."""
        if not pending:
            return None, 0
        if len(pending) > 12:
            raise RuntimeError("at most HS_ASM_MAX_JOBS jobs ride in one launch")
        return (hsAsmJob * len(pending))(*[d[0] for d in pending]), len(pending)

    # ---- per-ray sampler kernels (include/holoscene_hip.h section 3)
    @staticmethod
    def sampler_update(z, sdf, m_old, samples, new_sdf, beta, beta0, eps, beta_iters, beta_max, gate=None, m_dev=None):
        lib = load_library()
        R, ld = z.shape
        _check(lib.hs_sampler_update(_dev(z, "z"), _dev(sdf, "sdf"), ld, m_old, _dev(samples, "samples"), _dev(new_sdf, "new_sdf"),
                                     samples.shape[1], _dev(beta, "beta"), _dev(beta0, "beta0"), ctypes.c_float(eps), beta_iters,
                                     _dev(beta_max, "beta_max"), R, ctypes.byref(_gate(gate)), _dev(m_dev, "m_dev", torch.int32), _stream()),
               "hs_sampler_update")

    @staticmethod
    def sampler_update_draw(z, sdf, m_old, samples, new_sdf, beta, beta0, eps, beta_iters, beta_max, gate, add_tiny, out, cam_loc, ray_dirs,
                            divide_factor, x, x01):
        """hs_sampler_update with the next round's draw (and its positions) fused in; out [R, n_out]."""
        lib = load_library()
        R, ld = z.shape
        _check(lib.hs_sampler_update_draw(_dev(z, "z"), _dev(sdf, "sdf"), ld, m_old, _dev(samples, "samples"), _dev(new_sdf, "new_sdf"),
                                          samples.shape[1], _dev(beta, "beta"), _dev(beta0, "beta0"), ctypes.c_float(eps), beta_iters,
                                          _dev(beta_max, "beta_max"), R, ctypes.byref(_gate(gate)), ctypes.c_float(add_tiny), out.shape[1],
                                          _dev(out, "out"), _dev(cam_loc, "cam_loc"), _dev(ray_dirs, "ray_dirs"), ctypes.c_float(divide_factor),
                                          _dev(x, "x"), _dev(x01, "x01"), _stream()), "hs_sampler_update_draw")

    @staticmethod
    def sampler_draw(z, sdf, m, beta, mode, add_tiny, u, n_out, out, gate=None, m_dev=None):
        lib = load_library()
        R, ld = z.shape
        _check(lib.hs_sampler_draw(_dev(z, "z"), _dev(sdf, "sdf"), ld, m, _dev(beta, "beta"), mode, ctypes.c_float(add_tiny),
                                   _dev(u, "u"), n_out, _dev(out, "out"), R, ctypes.byref(_gate(gate)), _dev(m_dev, "m_dev", torch.int32),
                                   _stream()), "hs_sampler_draw")

    @staticmethod
    def sampler_draw_step(z, sdf, beta, mode, add_tiny, u, n_out, out, ctl_in, ctl_out, beta_max, beta0, s_new, max_rounds,
                          cam_loc=None, ray_dirs=None, divide_factor=1.0, x=None, x01=None):
        """ctl_in / ctl_out: two different float32[4] hsSamplerCtl slots (see sampler_step)."""
        lib = load_library()
        R, ld = z.shape
        _check(lib.hs_sampler_draw_step(_dev(z, "z"), _dev(sdf, "sdf"), ld, _dev(beta, "beta"), mode, ctypes.c_float(add_tiny), _dev(u, "u"), n_out,
                                        _dev(out, "out"), R, _dev(ctl_in, "ctl_in"), _dev(ctl_out, "ctl_out"), _dev(beta_max, "beta_max"),
                                        _dev(beta0, "beta0"), s_new, max_rounds, _dev(cam_loc, "cam_loc"), _dev(ray_dirs, "ray_dirs"),
                                        ctypes.c_float(divide_factor), _dev(x, "x"), _dev(x01, "x01"), _stream()), "hs_sampler_draw_step")

    @staticmethod
    def sampler_draw_steps(z, sdf, beta, mode, add_tiny, u, n_out, out, ctl_in, ctl_out, beta_max, beta0, s_new, max_rounds, n_steps):
        """sampler_draw_step with the step rule applied n_steps times on beta_max[0 .. n_steps) (the state after all rounds in one go)."""
        lib = load_library()
        R, ld = z.shape
        _check(lib.hs_sampler_draw_steps(_dev(z, "z"), _dev(sdf, "sdf"), ld, _dev(beta, "beta"), mode, ctypes.c_float(add_tiny), _dev(u, "u"), n_out,
                                         _dev(out, "out"), R, _dev(ctl_in, "ctl_in"), _dev(ctl_out, "ctl_out"), _dev(beta_max, "beta_max"),
                                         _dev(beta0, "beta0"), s_new, max_rounds, n_steps, None, None, ctypes.c_float(1.0), None, None, _stream()),
               "hs_sampler_draw_steps")

    @staticmethod
    def sampler_tail(z, sdf, beta, add_tiny, u, n_out, out, ctl_in, ctl_out, beta_max, beta0, s_new, max_rounds, n_steps, u_pick, pick_in, n_extra,
                     near, far, eik_idx, z_out, z_eik, near_rays=None, far_rays=None, eik_u=None):
        """sampler_draw_steps (final draw) + sampler_pick + sampler_final in one launch (hs_sampler_tail)."""
        lib = load_library()
        R, ld = z.shape
        _check(lib.hs_sampler_tail(_dev(z, "z"), _dev(sdf, "sdf"), ld, _dev(beta, "beta"), ctypes.c_float(add_tiny), _dev(u, "u"), n_out, _dev(out, "out"), R,
                                   _dev(ctl_in, "ctl_in"), _dev(ctl_out, "ctl_out"), _dev(beta_max, "beta_max"), _dev(beta0, "beta0"), s_new, max_rounds,
                                   n_steps, _dev(u_pick, "u_pick"), _dev(pick_in, "pick", torch.int64), int(n_extra), ctypes.c_float(near),
                                   ctypes.c_float(far), _dev(near_rays, "near_rays"), _dev(far_rays, "far_rays"), _dev(eik_idx, "eik_idx", torch.int64),
                                   _dev(eik_u, "eik_u"), _dev(z_out, "z_out"), _dev(z_eik, "z_eik"), _stream()), "hs_sampler_tail")

    @staticmethod
    def sampler_step(ctl, beta_max, beta0, s_new, max_rounds):
        """ctl: float32[4] device tensor holding an hsSamplerCtl {running, half, m (int32 bits), rounds (int32 bits)}."""
        lib = load_library()
        _check(lib.hs_sampler_step(_dev(ctl, "ctl"), _dev(beta_max, "beta_max"), _dev(beta0, "beta0"), s_new, max_rounds, _stream()),
               "hs_sampler_step")

    @staticmethod
    def sampler_pick(ctl, u, n_extra, pick):
        lib = load_library()
        _check(lib.hs_sampler_pick(_dev(ctl, "ctl"), _dev(u, "u"), n_extra, _dev(pick, "pick", torch.int64), _stream()), "hs_sampler_pick")

    @staticmethod
    def sampler_final(z_samples, z, pick, near, far, eik_idx, z_out, z_eik, near_rays=None, far_rays=None, eik_u=None):
        lib = load_library()
        R, ld = z.shape
        n_extra = 0 if pick is None else pick.numel()
        _check(lib.hs_sampler_final(_dev(z_samples, "z_samples"), z_samples.shape[1], _dev(z, "z"), ld, _dev(pick, "pick", torch.int64), n_extra,
                                    ctypes.c_float(near), ctypes.c_float(far), _dev(eik_idx, "eik_idx", torch.int64), _dev(z_out, "z_out"),
                                    _dev(z_eik, "z_eik"), R, _dev(near_rays, "near_rays"), _dev(far_rays, "far_rays"), _dev(eik_u, "eik_u"), _stream()),
               "hs_sampler_final")

    @staticmethod
    def ray_setup(uv, ray_offset, pose, intrinsics, t_rand, S, near, far_cap, bound, eps, ray_dirs, cam_loc, depth_scale, z0, beta_init,
                  divide_factor=1.0, x=None, x01=None, offset_shift=0.0, rot_out=None, beta_work=None, patch_u=None, patch=0):
        """patch_u (two U[0,1) draws on the device) with uv = None: the rays of the patch x patch pixel block the draws place (hs_ray_setup)."""
        lib = load_library()
        R = uv.shape[0] if uv is not None else int(patch) * int(patch)
        _check(lib.hs_ray_setup(_dev(uv, "uv"), _dev(ray_offset, "ray_offset"), _dev(pose, "pose"), _dev(intrinsics, "intrinsics"),
                                _dev(t_rand, "t_rand"), S, ctypes.c_float(near), ctypes.c_float(far_cap), ctypes.c_float(bound),
                                ctypes.c_float(eps), _dev(ray_dirs, "ray_dirs"), _dev(cam_loc, "cam_loc"), _dev(depth_scale, "depth_scale"),
                                _dev(z0, "z0"), _dev(beta_init, "beta_init"), R, ctypes.c_float(divide_factor), _dev(x, "x"),
                                _dev(x01, "x01"), ctypes.c_float(offset_shift), _dev(rot_out, "rot_out"), _dev(beta_work, "beta_work"),
                                _dev(patch_u, "patch_u"), int(patch), _stream()),
               "hs_ray_setup")

    # ---- value+Jacobian trunk elementwise stages (include/holoscene_hip.h section 4)
    @staticmethod
    def softplus_tangent_fwd(A, bias, out):
        lib = load_library()
        B, rows, W = A.shape
        dt = A.dtype
        _check(lib.hs_softplus_tangent_fwd(_dev(A, "A", dt), _dev(bias, "bias"), _dev(out, "out", dt), ctypes.c_int64(B), rows, W,
                                           _DTYPES[dt], _stream()), "hs_softplus_tangent_fwd")

    @staticmethod
    def softplus_tangent_bwd(A, bias, G, gA, gbias):
        lib = load_library()
        B, rows, W = A.shape
        dt = A.dtype
        _check(lib.hs_softplus_tangent_bwd(_dev(A, "A", dt), _dev(bias, "bias"), _dev(G, "G", dt), _dev(gA, "gA", dt), _dev(gbias, "gbias"),
                                           ctypes.c_int64(B), rows, W, _DTYPES[dt], _stream()), "hs_softplus_tangent_bwd")

    # ---- fused Adam (include/holoscene_hip.h section 5); `state` is a uint8 CUDA tensor holding an hsAdamState
    @staticmethod
    def adam_tick(state, beta1, beta2, gamma):
        lib = load_library()
        _check(lib.hs_adam_tick(_dev(state, "state", torch.uint8), ctypes.c_float(beta1), ctypes.c_float(beta2), ctypes.c_double(gamma),
                                _stream()), "hs_adam_tick")

    @staticmethod
    def adam_flat(p, g, m, v, begin, end, state, beta1, beta2, eps, grad_scale, g_base=0, mv_base=0):
        """g_base / mv_base: flat index of element 0 of a shard-local gradient / moment buffer (0 = full-length buffers)."""
        lib = load_library()
        _check(lib.hs_adam_flat_shard(_dev(p, "p"), _dev(g, "g"), _dev(m, "m"), _dev(v, "v"), ctypes.c_int64(begin), ctypes.c_int64(end),
                                      ctypes.c_int64(g_base), ctypes.c_int64(mv_base), _dev(state, "state", torch.uint8), ctypes.c_float(beta1),
                                      ctypes.c_float(beta2), ctypes.c_float(eps), ctypes.c_float(grad_scale), _stream()), "hs_adam_flat_shard")

    # ---- fused compositing (include/holoscene_hip.h section 6)
    @staticmethod
    def composite_fwd(z, sdf, raw, rgb, g, beta, depth_scale, sem_scale, weights, trans, rgb_out, depth_out, normal_out, sem_out, opac_out, rot=None):
        lib = load_library()
        R, N = z.shape
        K = raw.shape[-1]
        _check(lib.hs_composite_fwd(_dev(z, "z"), _dev(sdf, "sdf"), _dev(raw, "raw"), _dev(rgb, "rgb"), _dev(g, "g"), _dev(beta, "beta"),
                                    _dev(depth_scale, "depth_scale"), ctypes.c_float(sem_scale), R, N, K, _dev(weights, "weights"),
                                    _dev(trans, "trans"), _dev(rgb_out, "rgb_out"), _dev(depth_out, "depth_out"),
                                    _dev(normal_out, "normal_out"), _dev(sem_out, "sem_out"), _dev(opac_out, "opac_out"), _dev(rot, "rot"), _stream()),
               "hs_composite_fwd")

    @staticmethod
    def composite_bwd(z, sdf, raw, rgb, g, beta, depth_scale, sem_scale, g_w, g_rgb, g_depth, g_normal, g_sem, g_opac, d_sdf, d_raw, d_rgb,
                      d_g, d_beta, rot=None):
        lib = load_library()
        R, N = z.shape
        K = raw.shape[-1]
        _check(lib.hs_composite_bwd(_dev(z, "z"), _dev(sdf, "sdf"), _dev(raw, "raw"), _dev(rgb, "rgb"), _dev(g, "g"), _dev(beta, "beta"),
                                    _dev(depth_scale, "depth_scale"), ctypes.c_float(sem_scale), R, N, K, _dev(g_w, "g_w"),
                                    _dev(g_rgb, "g_rgb"), _dev(g_depth, "g_depth"), _dev(g_normal, "g_normal"), _dev(g_sem, "g_sem"),
                                    _dev(g_opac, "g_opac"), _dev(d_sdf, "d_sdf"), _dev(d_raw, "d_raw"), _dev(d_rgb, "d_rgb"),
                                    _dev(d_g, "d_g"), _dev(d_beta, "d_beta"), _dev(rot, "rot"), _stream()), "hs_composite_bwd")

    # ---- fused SDF-trunk inference (include/holoscene_hip.h section 7)
    @staticmethod
    def sdf_mlp_fwd(x, feat, W0, b0, W1, b1, W2, b2, d_out, select, out_min, out_raw, gate=None, feat_level_major=False):
        """select: -1 = min over all objects, k = object k, or a list / tuple of objects = min over that subset."""
        lib = load_library()
        bf = torch.bfloat16
        mask = 0
        if isinstance(select, (list, tuple)):
            if not select or min(select) < 0 or max(select) >= d_out:
                raise ValueError(f"object subset {select!r} outside [0, {d_out})")
            for k in select:
                mask |= 1 << int(k)
            select = -1
        _check(lib.hs_sdf_mlp_fwd(_dev(x, "x"), _dev(feat, "feat"), _dev(W0, "W0", bf), _dev(b0, "b0"), _dev(W1, "W1", bf), _dev(b1, "b1"),
                                  _dev(W2, "W2", bf), _dev(b2, "b2"), d_out, select, ctypes.c_uint64(mask), _dev(out_min, "out_min"), _dev(out_raw, "out_raw"),
                                  ctypes.c_int64(x.shape[0]), ctypes.byref(_gate(gate)), int(feat_level_major), _stream()), "hs_sdf_mlp_fwd")

    @staticmethod
    def sdf_mlp2_pack(W0, b0, W1, b1, W2, b2, d_out, log2_domain=True):
        """fp32 effective matrices -> (W0f, W1f, W2f, bias): the fragment-order bf16 images + bias block of the wave-tile kernels
        (log2_domain: csrc/sdf_mlp2.hip's scaled softplus; False: csrc/trunk_mlp2.hip, plain-domain activations)."""
        lib = load_library()
        lib.hs_sdf_mlp2_pack_bytes.restype = ctypes.c_int64
        dev = W0.device
        n = [int(lib.hs_sdf_mlp2_pack_bytes(i)) // 2 for i in range(3)]
        w12 = torch.empty(n[1] + n[2], device=dev, dtype=torch.bfloat16)      # W2f directly behind W1f: the kernel copies both in one sweep
        bufs = [torch.empty(n[0], device=dev, dtype=torch.bfloat16), w12[:n[1]], w12[n[1]:]]
        bias = torch.empty(int(lib.hs_sdf_mlp2_pack_bytes(3)) // 4, device=dev)
        if W0.stride(1) != 1 or W0.stride(0) < 71:
            raise RuntimeError("sdf_mlp2_pack: W0 must be row-major with at least 71 columns")
        _check(lib.hs_sdf_mlp2_pack(_dev(W0, "W0"), int(W0.stride(0)), _dev(b0, "b0"), _dev(W1, "W1"), _dev(b1, "b1"), _dev(W2, "W2"), _dev(b2, "b2"),
                                    d_out, *[_dev(b, "frag", torch.bfloat16) for b in bufs], _dev(bias, "bias"), int(log2_domain), _stream()), "hs_sdf_mlp2_pack")
        return (*bufs, bias)

    @staticmethod
    def sdf_mlp32_pack(W0, b0, W1, b1, W2, b2, d_out):
        """fp32 effective matrices -> (W0i, W1i, bias): the fp32 operand images of csrc/sdf_mlp32.hip (reduction order permuted to the
        accumulator layout, one block per 32-neuron tile)."""
        lib = load_library()
        lib.hs_sdf_mlp32_pack_bytes.restype = ctypes.c_int64
        dev = W0.device
        bufs = [torch.empty(int(lib.hs_sdf_mlp32_pack_bytes(i)) // 4, device=dev) for i in range(3)]
        if W0.stride(1) != 1 or W0.stride(0) < 71:
            raise RuntimeError("sdf_mlp32_pack: W0 must be row-major with at least 71 columns")
        _check(lib.hs_sdf_mlp32_pack(_dev(W0, "W0"), int(W0.stride(0)), _dev(b0, "b0"), _dev(W1, "W1"), _dev(b1, "b1"), _dev(W2, "W2"), _dev(b2, "b2"),
                                     int(d_out), *[_dev(b, "image") for b in bufs], _stream()), "hs_sdf_mlp32_pack")
        return tuple(bufs)

    @staticmethod
    def sdf_mlp32_fwd(x, feat, packed, d_out, select, out_min, out_raw, gate=None, feat_level_major=False):
        """hs_sdf_mlp32_fwd: the SDF trunk on fp32 operands (the reference's own arithmetic); arguments as sdf_mlp2_fwd, features fp32 only."""
        lib = load_library()
        mask = 0
        if isinstance(select, (list, tuple)):
            if not select or min(select) < 0 or max(select) >= d_out:
                raise ValueError(f"object subset {select!r} outside [0, {d_out})")
            for k in select:
                mask |= 1 << int(k)
            select = -1
        W0i, W1i, bias = packed
        _check(lib.hs_sdf_mlp32_fwd(_dev(x, "x"), _dev(feat, "feat"), _dev(W0i, "W0i"), _dev(W1i, "W1i"), _dev(bias, "bias"), int(d_out),
                                    int(select), ctypes.c_uint64(mask), _dev(out_min, "out_min"), _dev(out_raw, "out_raw"), ctypes.c_int64(x.shape[0]),
                                    ctypes.byref(_gate(gate)), int(bool(feat_level_major)), _stream()), "hs_sdf_mlp32_fwd")

    @staticmethod
    def sdf_mlp2_fwd(x, feat, packed, d_out, select, out_min, out_raw, gate=None, feat_level_major=False):
        lib = load_library()
        bf = torch.bfloat16
        mask = 0
        if isinstance(select, (list, tuple)):
            if not select or min(select) < 0 or max(select) >= d_out:
                raise ValueError(f"object subset {select!r} outside [0, {d_out})")
            for k in select:
                mask |= 1 << int(k)
            select = -1
        W0f, W1f, W2f, bias = packed
        # feat_level_major: False = fp32 [B, 32]; True = fp32 [16, B, 2]; 2 = int32 [16, B], a level's two channels as bf16 (fwd(out_bf16=True))
        _check(lib.hs_sdf_mlp2_fwd(_dev(x, "x"), _dev(feat, "feat", torch.int32 if int(feat_level_major) == 2 else torch.float32), _dev(W0f, "W0f", bf), _dev(W1f, "W1f", bf), _dev(W2f, "W2f", bf), _dev(bias, "bias"),
                                   d_out, select, ctypes.c_uint64(mask), _dev(out_min, "out_min"), _dev(out_raw, "out_raw"),
                                   ctypes.c_int64(x.shape[0]), ctypes.byref(_gate(gate)), int(feat_level_major), _stream()), "hs_sdf_mlp2_fwd")

    @staticmethod
    def sdf_sweep_fwd(x, x01, embeddings, offsets, S, H, packed, packed_b, d_out, select, out_min, out_raw, gate=None):
        """hs_sdf_sweep_fwd: hash gather + SDF trunk of one sampler sweep in one launch (16 levels x 2 channels; packed_b: the second output tile's pack
        for 33..64 outputs).  Bit-identical to fwd(out_bf16=True) + sdf_mlp2_fwd / _wide."""
        lib = load_library()
        bf = torch.bfloat16
        mask = 0
        if isinstance(select, (list, tuple)):
            if not select or min(select) < 0 or max(select) >= d_out:
                raise ValueError(f"object subset {select!r} outside [0, {d_out})")
            for k in select:
                mask |= 1 << int(k)
            select = -1
        if offsets.numel() != 17 or embeddings.dim() != 2 or embeddings.shape[1] != 2:
            raise RuntimeError("sdf_sweep_fwd: one table of 16 levels (offsets [17]) x 2 channels")
        W0f, W1f, W2f, bias = packed
        _check(lib.hs_sdf_sweep_fwd(_dev(x, "x"), _dev(x01, "x01"), _dev(embeddings, "embeddings"), _dev(offsets, "offsets", torch.int32), ctypes.c_float(S), int(H),
                                    _dev(W0f, "W0f", bf), _dev(W1f, "W1f", bf), _dev(W2f, "W2f", bf), _dev(bias, "bias"),
                                    _dev(packed_b[2], "W2f_b", bf) if packed_b is not None else None, _dev(packed_b[3], "bias_b") if packed_b is not None else None,
                                    d_out, select, ctypes.c_uint64(mask), _dev(out_min, "out_min"), _dev(out_raw, "out_raw"),
                                    ctypes.c_int64(x.shape[0]), ctypes.byref(_gate(gate)), _stream()), "hs_sdf_sweep_fwd")

    @staticmethod
    def sdf_mlp2_fwd_wide(x, feat, packed, packed_b, d_out, select, out_min, out_raw, gate=None, feat_level_major=False):
        """33 <= d_out <= 64 (hs_sdf_mlp2_fwd_wide): packed = sdf_mlp2_pack of rows 0..31 of the last layer, packed_b = of rows 32.. ."""
        lib = load_library()
        bf = torch.bfloat16
        mask = 0
        if isinstance(select, (list, tuple)):
            if not select or min(select) < 0 or max(select) >= d_out:
                raise ValueError(f"object subset {select!r} outside [0, {d_out})")
            for k in select:
                mask |= 1 << int(k)
            select = -1
        W0f, W1f, W2f, bias = packed
        _check(lib.hs_sdf_mlp2_fwd_wide(_dev(x, "x"), _dev(feat, "feat", torch.int32 if int(feat_level_major) == 2 else torch.float32), _dev(W0f, "W0f", bf),
                                        _dev(W1f, "W1f", bf), _dev(W2f, "W2f", bf), _dev(bias, "bias"), _dev(packed_b[2], "W2f_b", bf), _dev(packed_b[3], "bias_b"),
                                        d_out, select, ctypes.c_uint64(mask), _dev(out_min, "out_min"), _dev(out_raw, "out_raw"),
                                        ctypes.c_int64(x.shape[0]), ctypes.byref(_gate(gate)), int(feat_level_major), _stream()), "hs_sdf_mlp2_fwd_wide")

    @staticmethod
    def trunk_mlp2_fwd_wide(x, feat, dydx, packed, packed_b, d_out, H0, H1, Xp, jac_scale, split, ld=0, off=0, w2_planes=2):
        """33 <= d_out <= 64 (hs_trunk_mlp2_fwd_wide): packed = sdf_mlp2_pack(log2_domain=False) of the last layer's rows 0..31, packed_b = of rows
        32.. ; the split outputs only."""
        lib = load_library()
        bf = torch.bfloat16
        W0f, W1f, W2f, bias = packed
        n_main, sdf_raw, sdf, idx, grad, y_eik, min_eik, gtheta = split
        opt = lambda t, name, dt=torch.float32: _dev(t, name, dt).value if t is not None and t.numel() else None   # noqa: E731
        sp = hsTrunkSplit(int(n_main), opt(sdf_raw, "sdf_raw"), opt(sdf, "sdf"), _dev(idx, "idx", torch.int64).value, opt(grad, "grad"),
                          opt(y_eik, "y_eik"), opt(min_eik, "min_eik"), opt(gtheta, "grad_theta"))
        _check(lib.hs_trunk_mlp2_fwd_wide(_dev(x, "x"), _dev(feat, "feat"), _dev_at(dydx, "dydx", off * 6), _dev(W0f, "W0f", bf), _dev(W1f, "W1f", bf),
                                          _dev(W2f, "W2f", bf), _dev(bias, "bias"), _dev(packed_b[2], "W2f_b", bf), _dev(packed_b[3], "bias_b"), d_out,
                                          _dev(H0, "H0", bf), _dev(H1, "H1", bf), _dev(Xp, "Xp", bf), ctypes.c_int64(H0.shape[0]), ctypes.c_float(jac_scale),
                                          ctypes.byref(sp), ctypes.c_int64(ld), int(w2_planes), _stream()), "hs_trunk_mlp2_fwd_wide")

    @staticmethod
    def trunk_mlp2_fwd(x, feat, dydx, packed, d_out, H0, H1, Y, Xp, jac_scale, split=None, ld=0, off=0, w2_planes=2):
        """split = (n_main, sdf_raw, sdf, idx, grad, y_eik, min_eik, grad_theta): the kernel writes hs_trunk_split_fwd's outputs itself and
        Y (then None) is never stored.  ld / off: dydx is a [L, ld, 6] buffer whose points [off, off + M / 4) are this call's."""
        lib = load_library()
        bf = torch.bfloat16
        W0f, W1f, W2f, bias = packed
        sp = None
        if split is not None:
            n_main, sdf_raw, sdf, idx, grad, y_eik, min_eik, gtheta = split
            opt = lambda t, name, dt=torch.float32: _dev(t, name, dt).value if t is not None and t.numel() else None   # noqa: E731
            sp = hsTrunkSplit(int(n_main), opt(sdf_raw, "sdf_raw"), opt(sdf, "sdf"), _dev(idx, "idx", torch.int64).value, opt(grad, "grad"),
                              opt(y_eik, "y_eik"), opt(min_eik, "min_eik"), opt(gtheta, "grad_theta"))
        _check(lib.hs_trunk_mlp2_fwd(_dev(x, "x"), _dev(feat, "feat"), _dev_at(dydx, "dydx", off * 6), _dev(W0f, "W0f", bf), _dev(W1f, "W1f", bf), _dev(W2f, "W2f", bf),
                                     _dev(bias, "bias"), d_out, _dev(H0, "H0", bf), _dev(H1, "H1", bf), _dev(Y, "Y") if Y is not None else None,
                                     _dev(Xp, "Xp", bf), ctypes.c_int64(H0.shape[0]), ctypes.c_float(jac_scale),
                                     ctypes.byref(sp) if sp is not None else None, ctypes.c_int64(ld), int(w2_planes), _stream()), "hs_trunk_mlp2_fwd")

    @staticmethod
    def trunk_mlp2_columns():
        """Xp column of every reference input column (x | 6 octaves of sin, cos | 32 hash features): int64 [71]."""
        lib = load_library()
        return torch.tensor([int(lib.hs_trunk_mlp2_input_column(c)) for c in range(71)], dtype=torch.int64)

    @staticmethod
    def pack_bf16(jobs):
        """jobs: list of (src fp32 2-D tensor, dst bf16 2-D tensor, row0, col0, rows, cols, transpose[, scale]); rows/cols = valid
        extent in destination orientation, the rest of dst is zero-filled."""
        lib = load_library()
        arr = (hsPackJob * len(jobs))()
        for a, job in zip(arr, jobs):
            src, dst, row0, col0, rows, cols, tr = job[:7]
            a.scale = float(job[7]) if len(job) > 7 else 1.0
            a.src, a.dst = _dev(src, "src").value, _dev(dst, "dst", torch.bfloat16).value
            a.ld, a.row0, a.col0, a.rows, a.cols = src.shape[1], row0, col0, rows, cols
            a.dst_rows, a.dst_cols, a.transpose = dst.shape[0], dst.shape[1], int(tr)
        _check(lib.hs_pack_bf16(arr, len(jobs), _stream()), "hs_pack_bf16")

    @staticmethod
    def weight_norm_fwd(vs, gs):
        """[g * v / ||v||_row for v, g in zip(vs, gs)] in one launch (fp32)."""
        lib = load_library()
        arr = (hsWnJob * len(vs))()
        outs = []
        for a, v, g in zip(arr, vs, gs):
            W = torch.empty_like(v)
            a.v, a.g, a.W, a.rows, a.cols = _dev(v, "v").value, _dev(g, "g").value, _dev(W, "W").value, v.shape[0], v.shape[1]
            outs.append(W)
        _check(lib.hs_weight_norm(arr, len(vs), 0, _stream()), "hs_weight_norm")
        return outs

    @staticmethod
    def weight_norm_bwd(vs, gs, gWs):
        lib = load_library()
        arr = (hsWnJob * len(vs))()
        outs = []
        for a, v, g, gW in zip(arr, vs, gs, gWs):
            gv, gg = torch.empty_like(v), torch.empty_like(g)
            a.v, a.g, a.gW = _dev(v, "v").value, _dev(g, "g").value, _dev(gW, "gW").value
            a.gv, a.gg, a.rows, a.cols = _dev(gv, "gv").value, _dev(gg, "gg").value, v.shape[0], v.shape[1]
            outs += [gv, gg]
        _check(lib.hs_weight_norm(arr, len(vs), 1, _stream()), "hs_weight_norm")
        return outs

    @staticmethod
    def iter_prologue(vs, gs, rng_pool=None, rng_state=None, beta=None, beta_min=None, adam=None, zero=None, draw=None):
        """hs_iter_prologue: the weight-normalised matrices of (vs, gs), the pool of U[0, 1) draws, |beta| + beta_min and the optimiser tick in
        one launch.  adam: None or (state uint8 tensor, beta1, beta2, gamma).  draw: None or (draw_sched_plan(), n_uniform, total_pixels, n_out) --
        the iteration's batch drawn and gathered by the same launch (hs_iter_prologue_draw).  -> (Ws, beta_eff or None)"""
        lib = load_library()
        arr = (hsWnJob * max(len(vs), 1))()
        outs = []
        for a, v, g in zip(arr, vs, gs):
            W = torch.empty_like(v)
            a.v, a.g, a.W, a.rows, a.cols = _dev(v, "v").value, _dev(g, "g").value, _dev(W, "W").value, v.shape[0], v.shape[1]
            outs.append(W)
        beta_out = torch.empty_like(beta) if beta is not None else None
        st, b1, b2, gamma = adam if adam is not None else (None, 0.0, 0.0, 1.0)
        head = (arr, len(vs), _dev(rng_pool, "rng_pool"), ctypes.c_int64(0 if rng_pool is None else rng_pool.numel()),
                _dev(rng_state, "rng_state", torch.int64), _dev(beta, "beta"), _dev(beta_min, "beta_min"), _dev(beta_out, "beta_out"),
                0 if beta is None else beta.numel(), _dev(st, "adam state", torch.uint8), ctypes.c_float(b1), ctypes.c_float(b2),
                ctypes.c_double(gamma), _dev(zero, "zero"), ctypes.c_int64(0 if zero is None else zero.numel()))
        if draw is None:
            _check(lib.hs_iter_prologue(*head, _stream()), "hs_iter_prologue")
        else:
            (dst, darr, n_jobs, keep), n_uniform, total_pixels, n_out = draw
            _check(lib.hs_iter_prologue_draw(*head, ctypes.byref(dst), int(n_uniform), int(total_pixels), int(n_out), _dev(keep[3], "out", torch.int64), darr,
                                             n_jobs, _stream()), "hs_iter_prologue_draw")
        return outs, beta_out

    @staticmethod
    def iter_epilogue(vs, gs, gWs, outs, beta=None, g_beta_parts=(), g_beta_out=None):
        """hs_iter_epilogue: weight-norm backward of every layer + beta's backward in one launch.  outs: [(gv, gg)] destinations (e.g. views of
        the flat gradient buffer); g_beta_parts: up to four fp32 tensors of partial cotangents of |beta| + beta_min (all summed)."""
        lib = load_library()
        arr = (hsWnJob * max(len(vs), 1))()
        for a, v, g, gW, (gv, gg) in zip(arr, vs, gs, gWs, outs):
            a.v, a.g, a.gW = _dev(v, "v").value, _dev(g, "g").value, _dev(gW, "gW").value
            a.gv, a.gg, a.rows, a.cols = _dev(gv, "gv").value, _dev(gg, "gg").value, v.shape[0], v.shape[1]
        n_beta = 0 if beta is None else beta.numel()
        parts = [t for t in g_beta_parts if t is not None and t.numel() > 0]
        ptrs = (ctypes.c_void_p * max(len(parts), 1))(*[_dev(t, "g_beta part").value for t in parts])
        lens = (ctypes.c_int32 * max(len(parts), 1))(*[t.numel() // max(n_beta, 1) for t in parts])
        _check(lib.hs_iter_epilogue(arr, len(vs), _dev(beta, "beta"), ptrs, lens, len(parts), _dev(g_beta_out, "g_beta_out"), n_beta, _stream()),
               "hs_iter_epilogue")

    @staticmethod
    def gather_plan(jobs):
        """jobs: list of (src [rows, ...], dst [n, ...], idx int64 [n]) device tensors -> a reusable launch description of
        dst[i] = src[idx[i]] for all of them (hs_gather_rows).  The plan keeps the tensors alive; run it with gather_rows(plan)."""
        arr = (hsGatherJob * len(jobs))()
        for a, (src, dst, idx) in zip(arr, jobs):
            if not (src.is_cuda and dst.is_cuda and idx.is_cuda and src.is_contiguous() and dst.is_contiguous() and idx.is_contiguous()):
                raise RuntimeError("gather_plan: contiguous CUDA tensors expected")
            if idx.dtype != torch.int64 or src.dtype != dst.dtype:
                raise RuntimeError("gather_plan: idx must be int64 and src / dst of one dtype")
            row = (src.numel() // src.shape[0]) * src.element_size()
            if dst.numel() * dst.element_size() != idx.numel() * row or row % 4:
                raise RuntimeError("gather_plan: dst must hold idx.numel() rows of src's row size (a multiple of 4 bytes)")
            a.src, a.dst, a.idx, a.n, a.row_bytes = src.data_ptr(), dst.data_ptr(), idx.data_ptr(), idx.numel(), row
        return (arr, len(jobs), jobs)

    @staticmethod
    def gather_rows(plan):
        lib = load_library()
        _check(lib.hs_gather_rows(plan[0], plan[1], _stream()), "hs_gather_rows")

    @staticmethod
    def draw_pixels(class_ptr, class_pix, out_off, n_cls, per_class, n_bg, n_uniform, total_pixels, seed, counter, out, n_out=None, gather=None):
        """One batch's pixel indices (hs_draw_pixels; include/holoscene_hip.h): class_ptr / class_pix / out_off int32 device tensors, out
        int64 [>= n_out].  gather: a gather_plan() whose jobs indexed by `out` are served in the same launch (hs_draw_gather)."""
        lib = load_library()
        i32 = torch.int32
        n_out = int(out.numel() if n_out is None else n_out)
        args = (_dev(class_ptr, "class_ptr", i32), _dev(class_pix, "class_pix", i32), _dev(out_off, "out_off", i32), int(n_cls), int(per_class),
                int(n_bg), int(n_uniform), int(total_pixels), n_out, ctypes.c_uint64(int(seed) & (2 ** 64 - 1)), ctypes.c_uint64(int(counter)),
                _dev(out, "out", torch.int64))
        if gather is None:
            _check(lib.hs_draw_pixels(*args, _stream()), "hs_draw_pixels")
        else:
            _check(lib.hs_draw_gather(*args, gather[0], gather[1], _stream()), "hs_draw_gather")

    @staticmethod
    def draw_sched_plan(frames, jobs, out, sched, cursor, seed, counter_base):
        """A reusable description of hs_draw_gather_sched (the batch draw as a node of the iteration's graph; include/holoscene_hip.h: hsDrawSched).
        frames: per frame (class_ptr, class_pix, out_off, n_cls, per_class, n_bg) as draw_pixels takes them; jobs: (src, dst, idx) with src a
        tensor or a LIST of per-frame tensors, idx the drawn-index tensor `out`, another int64 index tensor, or None (= the frame's own row);
        sched int32 [n_sched], cursor int64 [2] device tensors.  The plan keeps every tensor alive."""
        if len(jobs) > GATHER_MAX_JOBS:
            raise RuntimeError("draw_sched_plan: too many gather jobs")
        dev = out.device
        arr = (hsGatherJob * len(jobs))()
        descs = (hsFrameDesc * len(frames))()
        for f, (ptr, pix, off, n_cls, per_class, n_bg) in enumerate(frames):
            d = descs[f]
            d.class_ptr, d.class_pix, d.out_off = _dev(ptr, "class_ptr", torch.int32).value, _dev(pix, "class_pix", torch.int32).value, _dev(off, "out_off", torch.int32).value
            d.n_cls, d.per_class, d.n_bg = int(n_cls), int(per_class), int(n_bg)
        for q, (a, (src, dst, idx)) in enumerate(zip(arr, jobs)):
            per_frame = isinstance(src, (list, tuple))
            srcs = list(src) if per_frame else [src]
            if per_frame and len(srcs) != len(frames):
                raise RuntimeError("draw_sched_plan: one source per frame expected")
            for t in srcs + [dst] + ([] if idx is None else [idx]):
                if not (t.is_cuda and t.is_contiguous()):
                    raise RuntimeError("draw_sched_plan: contiguous CUDA tensors expected")
            s0 = srcs[0]
            row = (s0.numel() // s0.shape[0]) * s0.element_size()
            n = dst.numel() * dst.element_size() // row
            if row % 4 or dst.numel() * dst.element_size() != n * row or any(t.dtype != dst.dtype or (t.numel() // t.shape[0]) * t.element_size() != row for t in srcs):
                raise RuntimeError("draw_sched_plan: dst must hold whole rows of src's row size (a multiple of 4 bytes), one dtype")
            if idx is not None and (idx.dtype != torch.int64 or idx.numel() != n):
                raise RuntimeError("draw_sched_plan: idx must be int64 with one entry per destination row")
            a.src = None if per_frame else s0.data_ptr()
            a.dst, a.idx, a.n, a.row_bytes = dst.data_ptr(), (None if idx is None else idx.data_ptr()), n, row
            if per_frame:
                for f, t in enumerate(srcs):
                    descs[f].src[q] = t.data_ptr()
        frames_dev = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(dev)
        st = hsDrawSched()
        st.frames, st.sched, st.cursor = frames_dev.data_ptr(), _dev(sched, "sched", torch.int32).value, _dev(cursor, "cursor", torch.int64).value
        st.seed, st.counter_base, st.n_sched, st.n_frames = int(seed) & (2 ** 64 - 1), int(counter_base) & (2 ** 64 - 1), sched.numel(), len(frames)
        return (st, arr, len(jobs), (frames_dev, frames, jobs, out, sched, cursor))

    @staticmethod
    def draw_gather_sched(plan, n_uniform, total_pixels, n_out):
        lib = load_library()
        st, arr, n_jobs, keep = plan
        _check(lib.hs_draw_gather_sched(ctypes.byref(st), int(n_uniform), int(total_pixels), int(n_out), _dev(keep[3], "out", torch.int64), arr, n_jobs, _stream()),
               "hs_draw_gather_sched")

    # ---- reverse-over-reverse trunk of the rendered samples (csrc/trunk_rr.hip, csrc/wgrad_pairs.hip)
    @staticmethod
    def tp_rows(n):
        """Rows a tile-packed activation tensor holds for n samples (whole 32-row tiles)."""
        return 32 * ((int(n) + 31) // 32)

    RR_GY_BLOCKS = 512

    @staticmethod
    def trunk_rr_gy(g_raw, g_sdf, idx, K, gy, gb2_part):
        """gy: bf16 [n, 32] (K <= 32) or two planes [2, n, 32] (33..64 objects: 0..31 | 32..63).  gb2_part: None or fp32 [RR_GY_BLOCKS, 32 or 64]
        (per-block column sums of gy)."""
        lib = load_library()
        KP = 32 if K <= 32 else 64
        if tuple(gy.shape[:-2]) != ((2,) if KP == 64 else ()) or gy.shape[-1] != 32:
            raise RuntimeError("trunk_rr_gy: gy must be [n, 32] for K <= 32, [2, n, 32] for 33..64 objects")
        if gb2_part is not None and tuple(gb2_part.shape) != (_HipBackend.RR_GY_BLOCKS, KP):
            raise RuntimeError(f"trunk_rr_gy: gb2_part must be [RR_GY_BLOCKS, {KP}]")
        _check(lib.hs_trunk_rr_gy(_dev(g_raw, "g_raw"), _dev(g_sdf, "g_sdf"), _dev(idx, "idx", torch.int64), int(K), _dev(gy, "gy", torch.bfloat16),
                                  _dev(gb2_part, "gb2_part"), ctypes.c_int64(gy.shape[-2]), _stream()), "hs_trunk_rr_gy")

    @staticmethod
    def trunk_rr_gy_split(g_raw, g_sdf, idx, K, gy, gb2_part, idx_e, g_yeik, g_mineik, g_theta, g_img):
        """trunk_rr_gy(...) and trunk_split_bwd(None, None, idx_e, None, g_yeik, g_mineik, g_theta, Be, 0, K, g_img) in one launch (g_img: bf16
        [4 Be, 32 or 64], the output-cotangent image of the Be Eikonal points' value+Jacobian rows)."""
        lib = load_library()
        KP = 32 if K <= 32 else 64
        if tuple(gy.shape[:-2]) != ((2,) if KP == 64 else ()) or gy.shape[-1] != 32:
            raise RuntimeError("trunk_rr_gy_split: gy must be [n, 32] for K <= 32, [2, n, 32] for 33..64 objects")
        if gb2_part is not None and tuple(gb2_part.shape) != (_HipBackend.RR_GY_BLOCKS, KP):
            raise RuntimeError(f"trunk_rr_gy_split: gb2_part must be [RR_GY_BLOCKS, {KP}]")
        Be = idx_e.shape[0]
        if g_img.dim() != 2 or g_img.shape[0] != 4 * Be or g_img.shape[1] not in (32, 64):
            raise RuntimeError("trunk_rr_gy_split: g_img must be [4 Be, 32 or 64]")
        _check(lib.hs_trunk_rr_gy_split(_dev(g_raw, "g_raw"), _dev(g_sdf, "g_sdf"), _dev(idx, "idx", torch.int64), int(K), _dev(gy, "gy", torch.bfloat16),
                                        _dev(gb2_part, "gb2_part"), ctypes.c_int64(gy.shape[-2]), _dev(idx_e, "idx_e", torch.int64), _dev(g_yeik, "g_y_eik"),
                                        _dev(g_mineik, "g_min_eik"), _dev(g_theta, "g_grad_theta"), ctypes.c_int64(Be), int(g_img.shape[1]),
                                        _dev(g_img, "g_img", torch.bfloat16), _stream()), "hs_trunk_rr_gy_split")

    @staticmethod
    def trunk_pack_wide(W0, b0, W1, b1, W2, b2, d_out):
        """33 <= d_out <= 64: the weight images of a reverse-over-reverse training pass with two output tiles -- (packed, packed_b, rr, W2Tf_b): `packed` /
        `rr` as trunk_pack_all of the last layer's rows 0..31 with rr's gather table W2tab grown to fp32 [64, 256] (rows 32.. from the second half),
        packed_b = sdf_mlp2_pack(log2_domain=False) of rows 32.. (its W2f / bias are what the kernels read), W2Tf_b = the W2^T image of rows 32.. .
        Three launches per parameter state (the one-launch pack of the iteration covers d_out <= 32 only)."""
        cls = _HipBackend
        c = lambda t: t.contiguous()  # noqa: E731
        packed, rr, _ = cls.trunk_pack_all(W0, b0, W1, b1, c(W2[:32]), c(b2[:32]), 32, transposes=False)
        packed_b = cls.sdf_mlp2_pack(W0, b0, W1, b1, c(W2[32:]), c(b2[32:]), int(d_out) - 32, log2_domain=False)
        _, _, W2Tf_b, tab_b = cls.trunk_rr_pack(W0, W1, c(W2[32:]), int(d_out) - 32)
        W2tab = torch.cat([rr[3].view(32, 256), tab_b.view(32, 256)]).contiguous().view(-1)
        return packed, packed_b, (rr[0], rr[1], rr[2], W2tab), W2Tf_b

    @staticmethod
    def trunk_rr_fwd_wide(x, feat, dydx, packed, packed_b, rr, d_out, H0t, H1t, Xp, sdf_raw, sdf, idx, onehot, U0t, V1t, V0t, grad, uxh, jac_scale, ld=0):
        """trunk_rr_fwd for 33..64 objects (k_rr_fwd<true>): rr's W2tab is the 64-row table of trunk_pack_wide, onehot two planes [2, n, 32]."""
        lib = load_library()
        bf = torch.bfloat16
        W0f, W1f, W2f, bias = packed
        W1Tf, W0Tf, _, W2tab = rr
        if W2tab.numel() != 64 * 256 or tuple(onehot.shape) != (2, x.shape[0], 32) or tuple(sdf_raw.shape) != (x.shape[0], d_out):
            raise RuntimeError("trunk_rr_fwd_wide: W2tab [64 * 256], onehot [2, n, 32], sdf_raw [n, d_out]")
        _check(lib.hs_trunk_rr_fwd_wide(_dev(x, "x"), _dev(feat, "feat"), _dev(dydx, "dydx"), _dev(W0f, "W0f", bf), _dev(W1f, "W1f", bf), _dev(W2f, "W2f", bf),
                                        _dev(bias, "bias"), _dev(packed_b[2], "W2f_b", bf), _dev(packed_b[3], "bias_b"), _dev(W2tab, "W2tab"),
                                        _dev(W1Tf, "W1Tf", bf), _dev(W0Tf, "W0Tf", bf), int(d_out), _dev(H0t, "H0t", bf), _dev(H1t, "H1t", bf), _dev(Xp, "Xp", bf),
                                        _dev(sdf_raw, "sdf_raw"), _dev(sdf, "sdf"), _dev(idx, "idx", torch.int64), _dev(onehot, "onehot", bf), _dev(U0t, "U0t", bf),
                                        _dev(V1t, "V1t", bf), _dev(V0t, "V0t", bf), _dev(grad, "grad"), _dev(uxh, "uxh"), ctypes.c_float(jac_scale),
                                        ctypes.c_int64(x.shape[0]), ctypes.c_int64(ld), _stream()), "hs_trunk_rr_fwd_wide")

    @staticmethod
    def trunk_rr_bwd_value_wide(gy, rr, W2Tf_b, H0t, H1t, A0pt, A1pt, A0t, A1t, g_feat, n, ld=0):
        """trunk_rr_bwd_value for 33..64 objects: gy two planes [2, n, 32]."""
        lib = load_library()
        bf = torch.bfloat16
        W1Tf, W0Tf, W2Tf, _ = rr
        if tuple(gy.shape) != (2, n, 32):
            raise RuntimeError("trunk_rr_bwd_value_wide: gy must be [2, n, 32]")
        _check(lib.hs_trunk_rr_bwd_value_wide(_dev(gy, "gy", bf), _dev(W2Tf, "W2Tf", bf), _dev(W2Tf_b, "W2Tf_b", bf), _dev(W1Tf, "W1Tf", bf), _dev(W0Tf, "W0Tf", bf),
                                              _dev(H0t, "H0t", bf), _dev(H1t, "H1t", bf), _dev(A0pt, "A0pt", bf), _dev(A1pt, "A1pt", bf), _dev(A0t, "A0t", bf),
                                              _dev(A1t, "A1t", bf), _dev(g_feat, "g_feat"), ctypes.c_int64(n), ctypes.c_int64(ld), _stream()),
               "hs_trunk_rr_bwd_value_wide")

    @staticmethod
    def trunk_rr_pack(W0, W1, W2, d_out):
        """fp32 effective matrices -> (W1Tf, W0Tf, W2Tf, W2tab): fragment images of the transposed matrices + the fp32 gather table of W2."""
        lib = load_library()
        lib.hs_trunk_rr_pack_bytes.restype = ctypes.c_int64
        dev, bf = W0.device, torch.bfloat16
        n = [int(lib.hs_trunk_rr_pack_bytes(i)) for i in range(4)]
        W1Tf, W0Tf, W2Tf = (torch.empty(n[i] // 2, device=dev, dtype=bf) for i in range(3))
        W2tab = torch.empty(n[3] // 4, device=dev)
        if W0.stride(1) != 1 or W0.stride(0) < 71:
            raise RuntimeError("trunk_rr_pack: W0 must be row-major with at least 71 columns")
        _check(lib.hs_trunk_rr_pack(_dev(W0, "W0"), int(W0.stride(0)), _dev(W1, "W1"), _dev(W2, "W2"), int(d_out), _dev(W1Tf, "W1Tf", bf),
                                    _dev(W0Tf, "W0Tf", bf), _dev(W2Tf, "W2Tf", bf), _dev(W2tab, "W2tab"), _stream()), "hs_trunk_rr_pack")
        return W1Tf, W0Tf, W2Tf, W2tab

    @staticmethod
    def trunk_pack_all(W0, b0, W1, b1, W2, b2, d_out, transposes):
        """sdf_mlp2_pack(log2_domain=False) + trunk_rr_pack (+ the row-major bf16 transposes w1t, w2t, w0t when `transposes`) in ONE launch
        -> (packed, rr, (w1t, w2t, w0t) or None)."""
        lib = load_library()
        lib.hs_sdf_mlp2_pack_bytes.restype = ctypes.c_int64
        lib.hs_trunk_rr_pack_bytes.restype = ctypes.c_int64
        dev, bf = W0.device, torch.bfloat16
        n = [int(lib.hs_sdf_mlp2_pack_bytes(i)) // 2 for i in range(3)]
        w12 = torch.empty(n[1] + n[2], device=dev, dtype=bf)
        packed = (torch.empty(n[0], device=dev, dtype=bf), w12[:n[1]], w12[n[1]:], torch.empty(int(lib.hs_sdf_mlp2_pack_bytes(3)) // 4, device=dev))
        m = [int(lib.hs_trunk_rr_pack_bytes(i)) for i in range(4)]
        rr = tuple(torch.empty(m[i] // 2, device=dev, dtype=bf) for i in range(3)) + (torch.empty(m[3] // 4, device=dev),)
        tr = (torch.empty(256, 256, device=dev, dtype=bf), torch.empty(256, 32, device=dev, dtype=bf), torch.empty(256, 256, device=dev, dtype=bf)) if transposes else None
        if W0.stride(1) != 1 or W0.stride(0) < 71:
            raise RuntimeError("trunk_pack_all: W0 must be row-major with at least 71 columns")
        _check(lib.hs_trunk_pack_all(_dev(W0, "W0"), int(W0.stride(0)), int(W0.shape[1]), _dev(b0, "b0"), _dev(W1, "W1"), _dev(b1, "b1"), _dev(W2, "W2"),
                                     _dev(b2, "b2"), int(d_out), *[_dev(t, "frag", bf) for t in packed[:3]], _dev(packed[3], "bias"),
                                     *[_dev(t, "frag", bf) for t in rr[:3]], _dev(rr[3], "W2tab"),
                                     *([_dev(t, "transpose", bf) for t in tr] if tr else [None, None, None]), _stream()), "hs_trunk_pack_all")
        return packed, rr, tr

    @staticmethod
    def pack_iteration(trunk=None, colour=None):
        """Every weight image of a Stage-1 iteration in one launch (hs_pack_iteration).
        trunk: None or (W0, b0, W1, b1, W2, b2, d_out, sampler, training, transposes) -- fp32 effective matrices; the three flags say which
        image families to build; colour: None or ((Wc0, Wc1, Wr0, Wr1, Wr2), (bc0, bc1, br0, br1, br2), transposed).
        -> {"sdf": sdf_mlp2_pack's tuple | None, "trunk": trunk_pack_all's (packed, rr, tr) | None, "appear": appearance2_pack's dict | None}"""
        lib = load_library()
        for fn in (lib.hs_sdf_mlp2_pack_bytes, lib.hs_trunk_rr_pack_bytes, lib.hs_appearance2_pack_bytes, lib.hs_appearance2_pack_t_bytes):
            fn.restype = ctypes.c_int64
        bf, u8 = torch.bfloat16, torch.uint8
        out = {"sdf": None, "trunk": None, "appear": None}
        targs = [None, 0, 0, None, None, None, None, None, 0] + [None] * 15
        keep = []
        if trunk is not None:
            W0, b0, W1, b1, W2, b2, d_out, sampler, training, transposes = trunk
            dev = W0.device
            if W0.stride(1) != 1 or W0.stride(0) < 71:
                raise RuntimeError("pack_iteration: W0 must be row-major with at least 71 columns")
            n = [int(lib.hs_sdf_mlp2_pack_bytes(i)) // 2 for i in range(3)]
            nb = int(lib.hs_sdf_mlp2_pack_bytes(3)) // 4

            def frag_set():
                w12 = torch.empty(n[1] + n[2], device=dev, dtype=bf)      # W2f directly behind W1f: the kernels copy both in one sweep
                return (torch.empty(n[0], device=dev, dtype=bf), w12[:n[1]], w12[n[1]:], torch.empty(nb, device=dev))

            sp = frag_set() if sampler else (None,) * 4
            packed = rr = tr = None
            if training:
                packed = frag_set()
                m = [int(lib.hs_trunk_rr_pack_bytes(i)) for i in range(4)]
                rr = tuple(torch.empty(m[i] // 2, device=dev, dtype=bf) for i in range(3)) + (torch.empty(m[3] // 4, device=dev),)
                if transposes:
                    tr = (torch.empty(256, 256, device=dev, dtype=bf), torch.empty(256, 32, device=dev, dtype=bf), torch.empty(256, 256, device=dev, dtype=bf))
            pk, r_, t_ = packed or (None,) * 4, rr or (None,) * 4, tr or (None,) * 3
            targs = [_dev(W0, "W0"), int(W0.stride(0)), int(W0.shape[1]), _dev(b0, "b0"), _dev(W1, "W1"), _dev(b1, "b1"), _dev(W2, "W2"), _dev(b2, "b2"), int(d_out),
                     *[_dev(t, "frag", bf) for t in sp[:3]], _dev(sp[3], "bias"), *[_dev(t, "frag", bf) for t in pk[:3]], _dev(pk[3], "bias"),
                     *[_dev(t, "frag", bf) for t in r_[:3]], _dev(r_[3], "W2tab"), *[_dev(t, "transpose", bf) for t in t_]]
            out["sdf"] = sp if sampler else None
            out["trunk"] = (packed, rr, tr) if training else None
        cargs = [None, None, None, 337] + [None] * 11
        if colour is not None:
            mats, biases, transposed = colour
            dev = mats[0].device
            nbs = [int(lib.hs_appearance2_pack_bytes(i)) for i in range(3)]
            P = {"stream": torch.empty(nbs[0], device=dev, dtype=u8), "R2f": torch.empty(nbs[1], device=dev, dtype=u8), "bias": torch.empty(nbs[2] // 4, device=dev),
                 "streamT": torch.empty(int(lib.hs_appearance2_pack_t_bytes()), device=dev, dtype=u8) if transposed else None}
            keep = [t.detach().float().contiguous() for t in tuple(mats) + tuple(biases)]
            cargs = [*[_dev(t, "w") for t in keep[:2]], _dev(keep[2], "wr0"), keep[2].shape[1], _dev(keep[3], "wr1"), _dev(keep[4], "wr2"),
                     *[_dev(t, "b") for t in keep[5:]], _dev(P["stream"], "stream", u8), _dev(P["R2f"], "R2f", u8), _dev(P["bias"], "bias"),
                     _dev(P["streamT"], "streamT", u8)]
            out["appear"] = P
        _check(lib.hs_pack_iteration(*targs, *cargs, _stream()), "hs_pack_iteration")
        return out

    @staticmethod
    def trunk_rr_fwd_value(x, feat, packed, d_out, H0t, H1t, Xp, sdf_raw, sdf, idx, onehot):
        lib = load_library()
        bf = torch.bfloat16
        W0f, W1f, W2f, bias = packed
        _check(lib.hs_trunk_rr_fwd_value(_dev(x, "x"), _dev(feat, "feat"), _dev(W0f, "W0f", bf), _dev(W1f, "W1f", bf), _dev(W2f, "W2f", bf), _dev(bias, "bias"),
                                         int(d_out), _dev(H0t, "H0t", bf), _dev(H1t, "H1t", bf), _dev(Xp, "Xp", bf), _dev(sdf_raw, "sdf_raw"), _dev(sdf, "sdf"),
                                         _dev(idx, "idx", torch.int64), _dev(onehot, "onehot", bf), ctypes.c_int64(x.shape[0]), _stream()), "hs_trunk_rr_fwd_value")

    @staticmethod
    def trunk_rr_fwd_grad(x, dydx, idx, rr, H0t, H1t, U0t, V1t, V0t, grad, uxh, jac_scale, ld=0):
        lib = load_library()
        bf = torch.bfloat16
        W1Tf, W0Tf, _, W2tab = rr
        _check(lib.hs_trunk_rr_fwd_grad(_dev(x, "x"), _dev(dydx, "dydx"), _dev(idx, "idx", torch.int64), _dev(W2tab, "W2tab"), _dev(W1Tf, "W1Tf", bf),
                                        _dev(W0Tf, "W0Tf", bf), _dev(H0t, "H0t", bf), _dev(H1t, "H1t", bf), _dev(U0t, "U0t", bf), _dev(V1t, "V1t", bf),
                                        _dev(V0t, "V0t", bf), _dev(grad, "grad"), _dev(uxh, "uxh"), ctypes.c_float(jac_scale), ctypes.c_int64(x.shape[0]),
                                        ctypes.c_int64(ld), _stream()), "hs_trunk_rr_fwd_grad")

    @staticmethod
    def trunk_rr_fwd(x, feat, dydx, packed, rr, d_out, H0t, H1t, Xp, sdf_raw, sdf, idx, onehot, U0t, V1t, V0t, grad, uxh, jac_scale, ld=0):
        """trunk_rr_fwd_value + trunk_rr_fwd_grad in one launch (csrc/trunk_rr.hip: k_rr_fwd)."""
        lib = load_library()
        bf = torch.bfloat16
        W0f, W1f, W2f, bias = packed
        W1Tf, W0Tf, _, W2tab = rr
        _check(lib.hs_trunk_rr_fwd(_dev(x, "x"), _dev(feat, "feat"), _dev(dydx, "dydx"), _dev(W0f, "W0f", bf), _dev(W1f, "W1f", bf), _dev(W2f, "W2f", bf),
                                   _dev(bias, "bias"), _dev(W2tab, "W2tab"), _dev(W1Tf, "W1Tf", bf), _dev(W0Tf, "W0Tf", bf), int(d_out), _dev(H0t, "H0t", bf),
                                   _dev(H1t, "H1t", bf), _dev(Xp, "Xp", bf), _dev(sdf_raw, "sdf_raw"), _dev(sdf, "sdf"), _dev(idx, "idx", torch.int64),
                                   _dev(onehot, "onehot", bf), _dev(U0t, "U0t", bf), _dev(V1t, "V1t", bf), _dev(V0t, "V0t", bf), _dev(grad, "grad"),
                                   _dev(uxh, "uxh"), ctypes.c_float(jac_scale), ctypes.c_int64(x.shape[0]), ctypes.c_int64(ld), _stream()), "hs_trunk_rr_fwd")

    @staticmethod
    def trunk_rr_bwd_grad(x, dydx, g_grad, uxh, idx, rr, packed, H0t, H1t, U0t, U0bt, A0pt, A1pt, U1bt, UXb, g_dydx, jac_scale, ld=0):
        lib = load_library()
        bf = torch.bfloat16
        W0f, W1f, _, _ = packed
        _check(lib.hs_trunk_rr_bwd_grad(_dev(x, "x"), _dev(dydx, "dydx"), _dev(g_grad, "g_grad"), _dev(uxh, "uxh"), _dev(idx, "idx", torch.int64),
                                        _dev(rr[3], "W2tab"), _dev(W0f, "W0f", bf), _dev(W1f, "W1f", bf), _dev(H0t, "H0t", bf), _dev(H1t, "H1t", bf),
                                        _dev(U0t, "U0t", bf), _dev(U0bt, "U0bt", bf), _dev(A0pt, "A0pt", bf), _dev(A1pt, "A1pt", bf), _dev(U1bt, "U1bt", bf),
                                        _dev(UXb, "UXb", bf), _dev(g_dydx, "g_dydx"), ctypes.c_float(jac_scale), ctypes.c_int64(x.shape[0]), ctypes.c_int64(ld),
                                        _stream()), "hs_trunk_rr_bwd_grad")

    @staticmethod
    def trunk_rr_bwd_value(gy, rr, H0t, H1t, A0pt, A1pt, A0t, A1t, g_feat, n, ld=0):
        lib = load_library()
        bf = torch.bfloat16
        W1Tf, W0Tf, W2Tf, _ = rr
        _check(lib.hs_trunk_rr_bwd_value(_dev(gy, "gy", bf), _dev(W2Tf, "W2Tf", bf), _dev(W1Tf, "W1Tf", bf), _dev(W0Tf, "W0Tf", bf), _dev(H0t, "H0t", bf),
                                         _dev(H1t, "H1t", bf), _dev(A0pt, "A0pt", bf), _dev(A1pt, "A1pt", bf), _dev(A0t, "A0t", bf), _dev(A1t, "A1t", bf),
                                         _dev(g_feat, "g_feat"), ctypes.c_int64(n), ctypes.c_int64(ld), _stream()), "hs_trunk_rr_bwd_value")

    WGP_KINDS = {(256, 256): 0, (256, 80): 1, (32, 256): 2, (256, 256, "rm"): 3, (256, 80, "rm"): 4, (256, 128, "rm"): 5, (32, 256, "rm"): 6, (256, 128, "tp"): 7}

    @staticmethod
    def wgrad_pairs(jobs, n, outs_into=None, colsum_out=None):
        """jobs: [((NA, W[, "rm"][, "ones"]), slices, (A0, B0), (A1, B1) or None[, rows])] -> bf16 partial stacks [slices, NA, MB] (MB = 128 for W = 80;
        "ones": column 80 of the result = column sums of A0, B0 may then be None), all
        in one launch (csrc/wgrad_pairs.hip).  Tile-packed operands cover n samples; "rm" jobs (both operands row-major) name their own
        row count (a multiple of 32 * slices).  outs_into: optional per-job destination tensors (e.g. slices of one stack, so that several
        jobs' partials are summed together by one hs_sum_slices job).  A shape tagged "colsum" also yields the per-slice column sums of A0
        (fp32 [slices, NA], appended to the list colsum_out; None for the other jobs)."""
        lib = load_library()
        bf = torch.bfloat16
        arr = (hsWgradPairJob * len(jobs))()
        outs = []
        for a, job in zip(arr, jobs):
            shape, slices, p0, p1 = job[:4]
            rows = int(job[4]) if len(job) > 4 else int(n)
            M = _HipBackend.tp_rows(rows)
            NA, W = shape[:2]
            MB = 128 if W == 80 else W
            part = outs_into[len(outs)] if outs_into is not None and outs_into[len(outs)] is not None else torch.empty(slices, NA, MB, device=p0[0].device, dtype=bf)
            if tuple(part.shape) != (slices, NA, MB) or part.dtype != bf or not part.is_contiguous():
                raise RuntimeError("wgrad_pairs: destination must be a contiguous bf16 [slices, NA, MB] tensor")
            a.ones = int("ones" in shape)
            a.reserved = int("reg" in shape) + 2 * int("consecutive" in shape)       # (tests: the two forms of the row stream against each other)
            cs = torch.empty(slices, NA, device=p0[0].device, dtype=torch.float32) if "colsum" in shape else None      # per-slice column sums of A0
            a.colsum = None if cs is None else cs.data_ptr()
            if colsum_out is not None:
                colsum_out.append(cs)
            a.A0, a.B0 = _dev(p0[0], "A0", bf).value, (_dev(p0[1], "B0", bf).value if p0[1] is not None else None)
            a.A1, a.B1 = (_dev(p1[0], "A1", bf).value, _dev(p1[1], "B1", bf).value) if p1 is not None else (None, None)
            a.part, a.M, a.rows, a.kind, a.slices = part.data_ptr(), M, rows, _HipBackend.WGP_KINDS[tuple(t for t in shape if t not in ("ones", "colsum", "reg", "consecutive"))], int(slices)
            outs.append(part)
        _check(lib.hs_wgrad_pairs(arr, len(jobs), _stream()), "hs_wgrad_pairs")
        return outs

    WGRAD_SHAPES = ((256, 256), (256, 128), (32, 256))

    @staticmethod
    def assemble(jobs, defer=False):
        """jobs: [((rows, cols), [term, ...])] with term = (src fp32 or bf16 tensor, ld, col0 or int32 column-map tensor[, red, red_stride]) ->
        fp32 tensors [rows, cols] = the sums of the terms (a job may name its destination: ((rows, cols), terms, (matrix, col0)) writes that
        column window of an existing fp32 matrix and returns the matrix), all in one launch
        (hs_assemble, csrc/small_ops.hip).  The element (r, c) of a term is src.flat[r * ld + col(c) (+ k * red_stride, summed over k < red)];
        a bf16 source is a stack of split-M weight-gradient partials [red, rows, ld]: its slice sum happens here too.
        defer: -> (results, pending): nothing is launched; `pending` goes to bwd(sums=) / bwd_jac(sums=) -- the jobs then ride in front of that table
        scatter's workgroups -- or to assemble_launch() (at most 12 jobs)."""
        lib = load_library()
        outs, keep = [], []
        if defer and len(jobs) > 12:
            raise RuntimeError("assemble(defer=True): at most HS_ASM_MAX_JOBS jobs")
        for k0 in range(0, len(jobs), 12):
            grp = jobs[k0:k0 + 12]
            arr = (hsAsmJob * len(grp))()
            for a, job in zip(arr, grp):
                (rows, cols), terms = job[:2]
                dev = terms[0][0].device
                if len(job) > 2 and job[2][1] is None:      # (fp32 contiguous tensor of rows * cols elements, None): the whole result goes there
                    out = job[2][0]
                    if out.dtype != torch.float32 or not out.is_contiguous() or out.numel() != rows * cols or not out.is_cuda:
                        raise RuntimeError("assemble: destination tensor must hold rows * cols contiguous fp32 values")
                    a.dst, a.dst_ld = out.data_ptr(), cols
                elif len(job) > 2:        # (tensor [rows, >= col0 + cols] fp32 contiguous, col0): a column window of an existing matrix
                    out, c0 = job[2]
                    if out.dtype != torch.float32 or not out.is_contiguous() or out.dim() != 2 or out.shape[0] != rows or c0 + cols > out.shape[1]:
                        raise RuntimeError("assemble: destination window outside its matrix")
                    a.dst, a.dst_ld = out.data_ptr() + 4 * int(c0), out.shape[1]
                else:
                    out = torch.empty(rows, cols, device=dev, dtype=torch.float32)
                    a.dst, a.dst_ld = out.data_ptr(), cols
                a.rows, a.cols, a.n_terms = rows, cols, len(terms)
                for t, term in zip(a.term, terms):
                    src, ld, col = term[:3]
                    t.src, t.ld, t.src_bf16 = _dev(src, "assemble source", src.dtype if src.dtype == torch.bfloat16 else torch.float32).value, int(ld), int(src.dtype == torch.bfloat16)
                    if torch.is_tensor(col):
                        t.col_map, t.col0 = _dev(col, "column map", torch.int32).value, 0
                        keep.append(col)
                    else:
                        t.col_map, t.col0 = None, int(col)
                    t.red, t.red_stride = (int(term[3]), int(term[4])) if len(term) > 3 else (1, 0)
                outs.append(out)
                keep += [t[0] for t in terms] + [out]
            if defer:
                pending = []
                for a in arr:
                    c = hsAsmJob()
                    ctypes.memmove(ctypes.addressof(c), ctypes.addressof(a), ctypes.sizeof(hsAsmJob))
                    pending.append((c, keep))       # (sources, column maps and destinations stay alive until the launch)
                return outs, pending
            _check(lib.hs_assemble(arr, len(grp), _stream()), "hs_assemble")
        return outs

    @staticmethod
    def assemble_launch(pending):
        """Launch what assemble(defer=True) described and no scatter took along (nothing pending: no launch)."""
        if pending:
            arr, n = _HipBackend._asm_array(pending)
            _check(load_library().hs_assemble(arr, n, _stream()), "hs_assemble")
            pending.clear()

    @staticmethod
    def abs_shift(x, shift=None, gy=None):
        """forward (gy is None): |x| + shift[0]; backward: gy * sgn(x) (hs_abs_shift)."""
        lib = load_library()
        out = torch.empty_like(x)
        if gy is None:
            _check(lib.hs_abs_shift(_dev(x, "x"), _dev(shift, "shift"), _dev(out, "y"), None, None, x.numel(), _stream()), "hs_abs_shift")
        else:
            _check(lib.hs_abs_shift(_dev(x, "x"), None, None, _dev(gy, "gy"), _dev(out, "gx"), x.numel(), _stream()), "hs_abs_shift")
        return out

    @staticmethod
    def wgrad_rows(pairs, slices):
        """pairs: [(A [M, NA] bf16, B [M, MB] bf16)] with (NA, MB) in WGRAD_SHAPES and M % slices == 0 -> bf16 stacks [slices, NA, MB] of the
        per-slice products A^T . B, all in one launch (at most 8 pairs per launch)."""
        lib = load_library()
        bf = torch.bfloat16
        outs = []
        for i0 in range(0, len(pairs), 8):
            grp = pairs[i0:i0 + 8]
            arr = (hsWgradJob * len(grp))()
            for a, (A, B) in zip(arr, grp):
                part = torch.empty(slices, A.shape[1], B.shape[1], device=A.device, dtype=bf)
                a.A, a.B, a.part = _dev(A, "A", bf).value, _dev(B, "B", bf).value, part.data_ptr()
                a.M, a.NA, a.MB = A.shape[0], A.shape[1], B.shape[1]
                if B.shape[0] != A.shape[0]:
                    raise RuntimeError("wgrad_rows: operands disagree on the number of rows")
                outs.append(part)
            _check(lib.hs_wgrad_rows(arr, len(grp), slices, _stream()), "hs_wgrad_rows")
        return outs

    @staticmethod
    def copy_many(dst, src):
        """dst[i].copy_(src[i]) for lists of contiguous fp32 CUDA tensors of equal sizes, one launch per 64 pairs (hs_copy_many)."""
        lib = load_library()
        for k in range(0, len(dst), 64):
            d_, s_ = dst[k:k + 64], src[k:k + 64]
            arr = (hsCopyJob * len(d_))()
            for a, d, s in zip(arr, d_, s_):
                if d.numel() != s.numel():
                    raise RuntimeError("copy_many: size mismatch")
                a.src, a.dst, a.n = _dev(s, "src").value, _dev(d, "dst").value, d.numel()
            _check(lib.hs_copy_many(arr, len(d_), _stream()), "hs_copy_many")

    @staticmethod
    def sum_slices(partials):
        """partials: list of bf16 or fp32 tensors [S, ...]; returns the fp32 sums over dim 0, all in one launch."""
        lib = load_library()
        arr = (hsSumJob * len(partials))()
        outs = []
        for a, t in zip(arr, partials):
            out = torch.empty(t.shape[1:], device=t.device, dtype=torch.float32)
            a.src, a.dst, a.n, a.slices = _dev(t, "partials", t.dtype).value, _dev(out, "out").value, out.numel(), t.shape[0]
            if t.dtype not in (torch.bfloat16, torch.float32):
                raise RuntimeError("sum_slices: bf16 or fp32 slices expected")
            a.src_f32 = int(t.dtype == torch.float32)
            outs.append(out)
        _check(lib.hs_sum_slices(arr, len(partials), _stream()), "hs_sum_slices")
        return outs

    # ---- fp32 products as split-bf16 MFMA sums (csrc/gemm_split.hip)
    @staticmethod
    def gemm_split_nt(a, b, bias=None, planes=3):
        """a [M, K] . b [N, K]^T (+ bias [N]) -> fp32 [M, N]; fp32 operands, rows may be strided (stride(1) == 1)."""
        lib = load_library()
        if a.dtype != torch.float32 or b.dtype != torch.float32 or a.dim() != 2 or b.dim() != 2 or a.shape[1] != b.shape[1]:
            raise RuntimeError("gemm_split_nt: fp32 [M, K] x [N, K] expected")
        a = a if a.stride(1) == 1 or a.shape[1] == 1 else a.contiguous()
        b = b if b.stride(1) == 1 or b.shape[1] == 1 else b.contiguous()
        M, K, N = a.shape[0], a.shape[1], b.shape[0]
        out = torch.empty(M, N, device=a.device, dtype=torch.float32)
        _check(lib.hs_gemm_split_nt(ctypes.c_void_p(a.data_ptr()), ctypes.c_int64(max(a.stride(0), K)), ctypes.c_void_p(b.data_ptr()), ctypes.c_int64(max(b.stride(0), K)),
                                    _dev(out, "out"), ctypes.c_int64(N), _dev(bias, "bias"), ctypes.c_int64(M), int(N), int(K), int(planes), _stream()), "hs_gemm_split_nt")
        return out

    @staticmethod
    def gemm_split_tn(a, b, slices, planes=3):
        """per-slice a [m, N]^T . b [m, K] -> fp32 partials [slices, N, K] (sum over dim 0 = a^T b)."""
        lib = load_library()
        if a.dtype != torch.float32 or b.dtype != torch.float32 or a.dim() != 2 or b.dim() != 2 or a.shape[0] != b.shape[0]:
            raise RuntimeError("gemm_split_tn: fp32 [M, N], [M, K] expected")
        a = a if a.stride(1) == 1 else a.contiguous()
        b = b if b.stride(1) == 1 else b.contiguous()
        M, N, K = a.shape[0], a.shape[1], b.shape[1]
        out = torch.empty(slices, N, K, device=a.device, dtype=torch.float32)
        _check(lib.hs_gemm_split_tn(ctypes.c_void_p(a.data_ptr()), ctypes.c_int64(max(a.stride(0), N)), ctypes.c_void_p(b.data_ptr()), ctypes.c_int64(max(b.stride(0), K)),
                                    _dev(out, "out"), ctypes.c_int64(M), int(N), int(K), int(slices), int(planes), _stream()), "hs_gemm_split_tn")
        return out

    # ---- colour branch, wave-tile form (csrc/appearance2.hip)
    @staticmethod
    def appearance2_pack(wc0, wc1, wr0, wr1, wr2, biases, transposed=True):
        """fp32 effective matrices + (bc0, bc1, br0, br1, br2) -> dict of fragment images (stream, R2f, bias, streamT: the backward kernel's
        transposed image, None without `transposed`), one launch."""
        lib = load_library()
        lib.hs_appearance2_pack_bytes.restype = ctypes.c_int64
        dev = wc0.device
        nb = [int(lib.hs_appearance2_pack_bytes(i)) for i in range(3)]
        lib.hs_appearance2_pack_t_bytes.restype = ctypes.c_int64
        P = {"stream": torch.empty(nb[0], device=dev, dtype=torch.uint8), "R2f": torch.empty(nb[1], device=dev, dtype=torch.uint8),
             "bias": torch.empty(nb[2] // 4, device=dev),
             "streamT": torch.empty(int(lib.hs_appearance2_pack_t_bytes()), device=dev, dtype=torch.uint8) if transposed else None}
        keep = [t.detach().float().contiguous() for t in (wc0, wc1, wr0, wr1, wr2) + tuple(biases)]
        _check(lib.hs_appearance2_pack(*[_dev(t, "w") for t in keep[:2]], _dev(keep[2], "wr0"), keep[2].shape[1], _dev(keep[3], "wr1"), _dev(keep[4], "wr2"),
                                       *[_dev(t, "b") for t in keep[5:]], _dev(P["stream"], "stream", torch.uint8), _dev(P["R2f"], "R2f", torch.uint8),
                                       _dev(P["bias"], "bias"), _dev(P["streamT"], "streamT", torch.uint8), _stream()), "hs_appearance2_pack")
        return P

    @staticmethod
    def appearance2_bwd(g_rgb, rgb, normals, masks, streamT, gy, GR1t, GR0t, GFVt, GHCt, d_normals, g_featc, gb2, normals_add=False):
        lib = load_library()
        bf = torch.bfloat16
        _check(lib.hs_appearance2_bwd(_dev(g_rgb, "g_rgb"), _dev(rgb, "rgb"), _dev(normals, "normals"), _dev(masks, "masks", torch.int32),
                                      _dev(streamT, "streamT", torch.uint8), _dev(gy, "gy", bf), _dev(GR1t, "GR1t", bf), _dev(GR0t, "GR0t", bf),
                                      _dev(GFVt, "GFVt", bf), _dev(GHCt, "GHCt", bf), _dev(d_normals, "d_normals"), _dev(g_featc, "g_featc"),
                                      _dev(gb2, "gb2"), ctypes.c_int64(g_rgb.shape[0]), int(bool(normals_add)), _stream()), "hs_appearance2_bwd")

    @staticmethod
    def appearance2_fwd(featc, points, dirs, normals, P, XAt, HCt, FVt, R0t, R1t, masks, rgb):
        """featc: fp32 [16, n, 2], or int32 [16, n] bf16 words (fwd(out_bf16=True))."""
        lib = load_library()
        bf, u8 = torch.bfloat16, torch.uint8
        words = featc.dtype == torch.int32
        _check(lib.hs_appearance2_fwd(_dev(featc, "featc", torch.int32 if words else torch.float32), _dev(points, "points"), _dev(dirs, "dirs"), _dev(normals, "normals"),
                                      _dev(P["stream"], "stream", u8), _dev(P["R2f"], "R2f", u8), _dev(P["bias"], "bias"),
                                      _dev(XAt, "XAt", bf), _dev(HCt, "HCt", bf), _dev(FVt, "FVt", bf), _dev(R0t, "R0t", bf), _dev(R1t, "R1t", bf),
                                      _dev(masks, "masks", torch.int32), _dev(rgb, "rgb"), ctypes.c_int64(points.shape[0]), int(words), _stream()), "hs_appearance2_fwd")

    @staticmethod
    def appearance_mask_words(B):
        lib = load_library()
        lib.hs_appearance_mask_words.restype = ctypes.c_int64
        return int(lib.hs_appearance_mask_words(ctypes.c_int64(B)))

    @staticmethod
    def appearance_fwd(featc, points, dirs, normals, W, biases, xin, hc, fv, r0, r1, rgb, relu_masks=None):
        """W: dict of packed bf16 operands (Wc0, Wc1, Wr0f, Wr0p, Wr1, Wr2); biases: (bc0, bc1, br0, br1, br2) fp32.
        relu_masks: optional int64 [appearance_mask_words(B)] receiving the ReLU signs as wave ballots (for appearance_bwd)."""
        lib = load_library()
        bf = torch.bfloat16
        _check(lib.hs_appearance_fwd(_dev(featc, "featc"), _dev(points, "points"), _dev(dirs, "dirs"), _dev(normals, "normals"),
                                     *[_dev(W[k], k, bf) for k in ("Wc0", "Wc1", "Wr0f", "Wr0p", "Wr1", "Wr2")],
                                     *[_dev(b, "bias") for b in biases], _dev(xin, "xin", bf), _dev(hc, "hc", bf), _dev(fv, "fv", bf),
                                     _dev(r0, "r0", bf), _dev(r1, "r1", bf), _dev(rgb, "rgb"), ctypes.c_int64(points.shape[0]),
                                     _dev(relu_masks, "relu_masks", torch.int64), _stream()),
               "hs_appearance_fwd")

    @staticmethod
    def appearance_bwd(g_rgb, rgb, normals, r1, r0, hc, W, gy, gA_r1, gA_r0, g_fv, gA_hc, d_normals, g_featc, gbias, relu_masks=None):
        """relu_masks: appearance_fwd's ballots -- then r1, r0, hc are not read (may be None)."""
        lib = load_library()
        bf = torch.bfloat16
        _check(lib.hs_appearance_bwd(_dev(g_rgb, "g_rgb"), _dev(rgb, "rgb"), _dev(normals, "normals"), _dev(r1, "r1", bf), _dev(r0, "r0", bf),
                                     _dev(hc, "hc", bf), *[_dev(W[k], k, bf) for k in ("Wr2t", "Wr1t", "Wr0ft", "Wr0nt", "Wc1t", "Wc0t")],
                                     _dev(gy, "gy", bf), _dev(gA_r1, "gA_r1", bf), _dev(gA_r0, "gA_r0", bf), _dev(g_fv, "g_fv", bf),
                                     _dev(gA_hc, "gA_hc", bf), _dev(d_normals, "d_normals"), _dev(g_featc, "g_featc"), _dev(gbias, "gbias"),
                                     ctypes.c_int64(g_rgb.shape[0]), _dev(relu_masks, "relu_masks", torch.int64), _stream()), "hs_appearance_bwd")

    @staticmethod
    def render_points(cam_loc, ray_dirs, z_vals, z_eik, eik_uniform, eik_jitter, divide_factor, x, x01, dirs_flat, eik_scale=1.0, eik_shift=0.0):
        lib = load_library()
        R, N = z_vals.shape
        _check(lib.hs_render_points(_dev(cam_loc, "cam_loc"), _dev(ray_dirs, "ray_dirs"), _dev(z_vals, "z_vals"), _dev(z_eik, "z_eik"),
                                    _dev(eik_uniform, "eik_uniform"), _dev(eik_jitter, "eik_jitter"), ctypes.c_int64(R), N, ctypes.c_float(divide_factor),
                                    _dev(x, "x"), _dev(x01, "x01"), _dev(dirs_flat, "dirs_flat"), ctypes.c_float(eik_scale), ctypes.c_float(eik_shift),
                                    _stream()), "hs_render_points")

    @staticmethod
    def ray_points(cam_loc, ray_dirs, z, x, x01, divide_factor, gate=None):
        lib = load_library()
        _check(lib.hs_ray_points(_dev(cam_loc, "cam_loc"), _dev(ray_dirs, "ray_dirs"), _dev(z, "z"), _dev(x, "x"), _dev(x01, "x01"),
                                 ctypes.c_int64(z.shape[0]), z.shape[1], ctypes.c_float(divide_factor), ctypes.byref(_gate(gate)), _stream()),
               "hs_ray_points")

    @staticmethod
    def trunk_mlp_fwd(X, W0, b0, W1, b1, W2, b2, d_out, H0, H1, Y, x=None, feat=None, dydx=None, Xout=None, L=0, C=0, jac_scale=0.0):
        lib = load_library()
        bf = torch.bfloat16
        _check(lib.hs_trunk_mlp_fwd(_dev(X, "X", bf), _dev(W0, "W0", bf), _dev(b0, "b0"), _dev(W1, "W1", bf), _dev(b1, "b1"), _dev(W2, "W2", bf),
                                    _dev(b2, "b2"), d_out, _dev(H0, "H0", bf), _dev(H1, "H1", bf), _dev(Y, "Y"), ctypes.c_int64(Y.shape[0]),
                                    _dev(x, "x"), _dev(feat, "feat"), _dev(dydx, "dydx"), _dev(Xout, "Xout", bf), L, C, ctypes.c_float(jac_scale),
                                    _stream()), "hs_trunk_mlp_fwd")

    @staticmethod
    def trunk_bwd_parts(M):
        return int(load_library().hs_trunk_bwd_parts(ctypes.c_int64(M)))

    @staticmethod
    def trunk_mlp_bwd(g, H1, H0, W2t, W1t, gA1, gA0, gb1, gb0, W0t=None, g_feat=None, g_dydx=None, L=0, C=0, jac_scale=0.0, gb2=None,
                      dW2_part=None, ld=0, off=0):
        """dW2_part: optional fp32 [trunk_bwd_parts(M), g.shape[1], 256] receiving per-workgroup slices of g^T . H1.
        ld / off: g_feat / g_dydx are [L, ld, .] buffers whose points [off, off + M / 4) this call writes."""
        lib = load_library()
        bf = torch.bfloat16
        if dW2_part is not None and tuple(dW2_part.shape) != (_HipBackend.trunk_bwd_parts(g.shape[0]), g.shape[-1], 256):
            raise RuntimeError("trunk_mlp_bwd: dW2_part has the wrong shape")
        _check(lib.hs_trunk_mlp_bwd(_dev(g, "g", bf), g.shape[-1], _dev(H1, "H1", bf), _dev(H0, "H0", bf), _dev(W2t, "W2t", bf),
                                    _dev(W1t, "W1t", bf), _dev(gA1, "gA1", bf), _dev(gA0, "gA0", bf), _dev(gb1, "gb1"), _dev(gb0, "gb0"),
                                    _dev(W0t, "W0t", bf), _dev_at(g_feat, "g_feat", off * C), _dev_at(g_dydx, "g_dydx", off * 3 * C), L, C,
                                    ctypes.c_float(jac_scale), ctypes.c_int64(g.shape[0]), _dev(gb2, "gb2"), _dev(dW2_part, "dW2_part"),
                                    ctypes.c_int64(ld), _stream()), "hs_trunk_mlp_bwd")

    @staticmethod
    def trunk_split_fwd(Y, n_main, K, sdf_raw, sdf, idx, grad, y_eik, min_eik, grad_theta):
        lib = load_library()
        _check(lib.hs_trunk_split_fwd(_dev(Y, "Y"), ctypes.c_int64(Y.shape[0] // 4), ctypes.c_int64(n_main), K, _dev(sdf_raw, "sdf_raw"), _dev(sdf, "sdf"),
                                      _dev(idx, "idx", torch.int64), _dev(grad, "grad"), _dev(y_eik, "y_eik"), _dev(min_eik, "min_eik"),
                                      _dev(grad_theta, "grad_theta"), _stream()), "hs_trunk_split_fwd")

    @staticmethod
    def trunk_split_bwd(g_raw, g_sdf, idx, g_grad, g_yeik, g_mineik, g_theta, B, n_main, K, g):
        lib = load_library()
        _check(lib.hs_trunk_split_bwd(_dev(g_raw, "g_sdf_raw"), _dev(g_sdf, "g_sdf"), _dev(idx, "idx", torch.int64), _dev(g_grad, "g_grad"),
                                      _dev(g_yeik, "g_y_eik"), _dev(g_mineik, "g_min_eik"), _dev(g_theta, "g_grad_theta"), ctypes.c_int64(B),
                                      ctypes.c_int64(n_main), K, g.shape[-1], _dev(g, "g", torch.bfloat16), _stream()), "hs_trunk_split_bwd")

    @staticmethod
    def softplus_tangent_bwd_h(H, G, gA, gbias):
        """H, G, gA: [4*points, W] rows grouped by point (value, d/dx, d/dy, d/dz)."""
        lib = load_library()
        dt = H.dtype
        _check(lib.hs_softplus_tangent_bwd_h(_dev(H, "H", dt), _dev(G, "G", dt), _dev(gA, "gA", dt), _dev(gbias, "gbias"),
                                             ctypes.c_int64(H.shape[0] // 4), H.shape[-1], _DTYPES[dt], _stream()), "hs_softplus_tangent_bwd_h")

    # ---- fused network-input builders (include/holoscene_hip.h section 8)
    @staticmethod
    def trunk_input_fwd(x, feat, dydx, out, nfreq, L, C, jac_scale):
        lib = load_library()
        dt = out.dtype
        _check(lib.hs_trunk_input_fwd(_dev(x, "x"), _dev(feat, "feat"), _dev(dydx, "dydx"), _dev(out, "out", dt), ctypes.c_int64(x.shape[0]),
                                      nfreq, L, C, ctypes.c_float(jac_scale), out.shape[-1], _DTYPES[dt], _stream()), "hs_trunk_input_fwd")

    @staticmethod
    def trunk_input_bwd(G, g_feat, g_dydx, nfreq, L, C, jac_scale):
        lib = load_library()
        dt = G.dtype
        _check(lib.hs_trunk_input_bwd(_dev(G, "G", dt), _dev(g_feat, "g_feat"), _dev(g_dydx, "g_dydx"), ctypes.c_int64(G.shape[0]), nfreq, L, C,
                                      ctypes.c_float(jac_scale), G.shape[-1], _DTYPES[dt], _stream()), "hs_trunk_input_bwd")

    @staticmethod
    def render_input_fwd(points, dirs, normals, fv, out, nfreq):
        lib = load_library()
        dt = out.dtype
        _check(lib.hs_render_input_fwd(_dev(points, "points"), _dev(dirs, "view_dirs"), _dev(normals, "normals"), _dev(fv, "feature_vectors", dt),
                                       _dev(out, "out", dt), ctypes.c_int64(points.shape[0]), nfreq, fv.shape[1], _DTYPES[dt], _stream()),
               "hs_render_input_fwd")

    @staticmethod
    def render_input_bwd(G, normals, d_normals, d_fv, nfreq, Fv):
        lib = load_library()
        dt = G.dtype
        _check(lib.hs_render_input_bwd(_dev(G, "G", dt), _dev(normals, "normals"), _dev(d_normals, "d_normals"), _dev(d_fv, "d_fv", dt),
                                       ctypes.c_int64(G.shape[0]), nfreq, Fv, _DTYPES[dt], _stream()), "hs_render_input_bwd")

    # ---- fused objective (include/holoscene_hip.h section 9)
    @staticmethod
    def loss_rays(rgb, rgb_gt, depth, depth_gt, nmap, n_gt, gt_mask, sdf, opac, segs, weights, out5, g_rgb, g_depth, g_nmap, g_opac):
        lib = load_library()
        R, N = sdf.shape
        K = opac.shape[1]
        w = [ctypes.c_float(float(x)) for x in weights]
        scratch = torch.empty(2 * R, device=rgb.device)
        _check(lib.hs_loss_rays(_dev(rgb, "rgb"), _dev(rgb_gt, "rgb_gt"), _dev(depth, "depth"), _dev(depth_gt, "depth_gt"), _dev(nmap, "normal_map"),
                                _dev(n_gt, "normal_gt"), _dev(gt_mask, "gt_mask"), _dev(sdf, "sdf"), _dev(opac, "opacity"),
                                _dev(segs, "segs", torch.int64), R, N, K, *w, _dev(out5, "out5"), _dev(g_rgb, "g_rgb"), _dev(g_depth, "g_depth"),
                                _dev(g_nmap, "g_normal_map"), _dev(g_opac, "g_opacity"), _dev(scratch, "scratch"), _stream()), "hs_loss_rays")

    @staticmethod
    def loss_stage1(rgb, rgb_gt, depth, depth_gt, nmap, n_gt, gt_mask, sdf, opac, segs, g1, g2, weights7, out8, g_rgb, g_depth, g_nmap, g_opac, d_g1, d_g2):
        lib = load_library()
        R, N = sdf.shape
        K = opac.shape[1]
        w = (ctypes.c_float * 7)(*[float(x) for x in weights7])
        scratch = torch.empty(2 * R + 2 * 256, device=rgb.device)      # HS_LOSS_EIK_BLOCKS
        _check(lib.hs_loss_stage1(_dev(rgb, "rgb"), _dev(rgb_gt, "rgb_gt"), _dev(depth, "depth"), _dev(depth_gt, "depth_gt"), _dev(nmap, "normal_map"),
                                  _dev(n_gt, "normal_gt"), _dev(gt_mask, "gt_mask"), _dev(sdf, "sdf"), _dev(opac, "opacity"),
                                  _dev(segs, "segs", torch.int64), R, N, K, _dev(g1, "g1"), _dev(g2, "g2"), ctypes.c_int64(g1.shape[0]), w,
                                  _dev(out8, "out8"), _dev(g_rgb, "g_rgb"), _dev(g_depth, "g_depth"), _dev(g_nmap, "g_normal_map"),
                                  _dev(g_opac, "g_opacity"), _dev(d_g1, "d_g1"), _dev(d_g2, "d_g2"), _dev(scratch, "scratch"), _stream()),
               "hs_loss_stage1")

    @staticmethod
    def bg_smooth_loss(depth, normal, labels, side, out, g_depth, g_normal):
        lib = load_library()
        _check(lib.hs_bg_smooth_loss(_dev(depth, "depth"), _dev(normal, "normal"), _dev(labels, "labels", torch.int64), side, _dev(out, "out"),
                                     _dev(g_depth, "g_depth"), _dev(g_normal, "g_normal"), _stream()), "hs_bg_smooth_loss")

    @staticmethod
    def loss_eikonal(g1, g2, w_eik, w_smooth, acc2, d_g1, d_g2):
        lib = load_library()
        _check(lib.hs_loss_eikonal(_dev(g1, "g1"), _dev(g2, "g2"), ctypes.c_int64(g1.shape[0]), ctypes.c_float(w_eik), ctypes.c_float(w_smooth),
                                   _dev(acc2, "acc2"), _dev(d_g1, "d_g1"), _dev(d_g2, "d_g2"), _stream()), "hs_loss_eikonal")


class hsAdamState(ctypes.Structure):
    _fields_ = [("step", ctypes.c_int64), ("group_end", ctypes.c_int64 * 2), ("lr0", ctypes.c_float * 3), ("lr", ctypes.c_float * 3),
                ("step_size", ctypes.c_float * 3), ("bc2_sqrt", ctypes.c_float)]


_backend = _HipBackend()

__all__ = ["_backend", "load_library", "LIB_PATH"]
