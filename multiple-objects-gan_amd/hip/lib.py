"""ctypes binding of libmogan_hip.so (include/mogan_hip.h).

The product path has NO fallback: if the shared object is missing or a call is made without a
GPU tensor, this raises.  Importing the module is harmless on a CPU-only box (so that
`__graft_entry__.build()` and the CPU test-suite can import the package); the library is opened
on first use.
"""
import contextlib
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("MOGAN_LIB") or os.path.join(_HERE, "libmogan_hip.so")   # MOGAN_LIB: tools/lab variants

P, I, F, L, Z = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_longlong, ctypes.c_size_t

# name -> argtypes (all return int unless listed in _RESTYPE); mirrors include/mogan_hip.h
SIGNATURES = {
    "mogan_abi_version": [],
    "mogan_gemm_set_split_target": [I],
    "mogan_stream_set_split_target": [P, I],
    "mogan_gemm_debug_force": [I, I],
    "mogan_gemm_group_min_tiles": [I],
    "mogan_mfma_form": [],
    "mogan_gemm_tune_set": [I, I, I, I, I, I, I],
    "mogan_gemm_tune_clear": [],
    "mogan_reserve_streams": [ctypes.c_int],
    "mogan_prof_enable": [I],
    "mogan_prof_collect": [P, I],
    "mogan_prof_dump": [ctypes.c_char_p],
    "mogan_conv2d_out_dims": [I, I, I, I, I, I, I, I, P, P],
    "mogan_conv2d_fwd": [P, P, P] + [I] * 11 + [P, Z, P],
    "mogan_logits_head_fwd": [P, P, P, P, I, I, I, P],
    "mogan_logits_head_bwd": [P, P, P, P, P, P, P, I, I, I, I, P],
    "mogan_conv2d_lrelu_fwd": [P, P, P] + [I] * 10 + [F, P, Z, P],
    "mogan_conv2d_affine_fwd": [P, P, P, P, P] + [I] * 11 + [P, Z, P],
    "mogan_affine_relu_bwd_out": [P, P, P, P, I, I, I, P],
    "mogan_conv2d_dgrad": [P, P, P] + [I] * 11 + [P, Z, P],
    "mogan_conv2d_wgrad": [P, P, P] + [I] * 12 + [P, Z, P],
    "mogan_wino_prep_bytes": [I] * 12,
    "mogan_wino_prep_group": [I, P, P, P, P, P, P],
    "mogan_conv_prep_bytes": [I] * 12,
    "mogan_conv_prep_group": [I, P, P, P, P, P, P, P],
    "mogan_conv2d_fwd_wp": [P, P, P, P] + [I] * 11 + [P, Z, P],
    "mogan_conv2d_dgrad_wp": [P, P, P, P] + [I] * 11 + [P, Z, P],
    "mogan_pk_conv_eligible": [I] * 11,
    "mogan_pk_weight_bytes": [I] * 6,
    "mogan_pk_weight_pack": [P, P] + [I] * 8 + [P],
    "mogan_pk_weight_pack_both": [P, P, P] + [I] * 7 + [P],
    "mogan_conv2d_fwd_pk": [P, P, P] + [I] * 10 + [P, Z, P],
    "mogan_conv2d_dgrad_pk": [P, P, P] + [I] * 10 + [P, Z, P],
    "mogan_pk_debug_force": [I, I, I],
    "mogan_pk_wgrad_eligible": [I] * 10 + [Z],
    "mogan_conv2d_wgrad_pk": [P, P, P] + [I] * 11 + [P, Z, P],
    "mogan_pk_panel_bytes": [I, I, I],
    "mogan_deep_block_eligible": [I] * 12,
    "mogan_deep_conv_bn_act_fwd": [P] * 11 + [I] * 10 + [F, F, I, F, I, P, Z, P],
    "mogan_deep_conv_bn_act_bwd": [P] * 9 + [I, P] + [I] * 10 + [I, F, I, P, Z, P],
    "mogan_upconv3x3_ws_bytes": [I, I],
    "mogan_upconv3x3_fwd": [P, P, P, I, I, I, I, I, P, Z, P],
    "mogan_lstm_encoder_fwd": [P] * 11 + [I] * 6 + [P],
    "mogan_upconv3x3_k4": [P, P, I, I, P],
    "mogan_upconv3x3_k4_group": [I, P, P, P, P, P],
    "mogan_upconv3x3_dgrad": [P, P, P, I, I, I, I, I, P, Z, P],
    "mogan_upconv3x3_wgrad": [P, P, P, I, I, I, I, I, I, P, Z, P],
    "mogan_down2_sum": [P, P, I, I, I, P],
    "mogan_bmm": [P, P, P, I, I, I, I] + [L] * 9 + [I, P, Z, P],
    "mogan_bn_ws_bytes": [I, I, I],
    "mogan_bn_stats": [P, I, I, I, F, F, P, P, P, P, P, Z, P],
    "mogan_bn_running_update": [P, P, P, P, I, L, F, F, P],
    "mogan_bn_act_fwd": [P, P, P, P, P, P, P, I, I, I, I, F, P],
    "mogan_bn_act_grouped_eligible": [I, I, I, I],
    "mogan_bn_act_grouped_fwd": [P] * 8 + [I] * 5 + [F, F, F, P],
    "mogan_bn_act_grouped_bwd": [P] * 9 + [I] * 5 + [F, I, P],
    "mogan_bn_act_fwd_fused": [P, P, P, P, P, P, P, P, P, I, I, I, I, F, F, F, P, Z, P],
    "mogan_bn_act_bwd": [P, P, P, P, P, P, P, P, P, I, I, I, I, F, I, P, Z, P],
    "mogan_affine_act_fwd": [P, P, P, P, I, I, I, I, F, P],
    "mogan_affine_act_bwd": [P, P, P, P, P, I, I, I, I, F, P],
    "mogan_act_fwd": [P, P, I, I, I, I, F, P],
    "mogan_act_bwd": [P, P, P, I, I, I, I, F, P],
    "mogan_bias_add": [P, P, I, I, I, P],
    "mogan_bias_grad": [P, P, I, I, I, I, P],
    "mogan_add": [P, P, P, L, P],
    "mogan_group_sum": [P, P, L, I, P],
    "mogan_group_bcast": [P, P, L, I, P],
    "mogan_scale": [P, F, P, L, P],
    "mogan_softmax_fwd": [P, P, P, L, I, L, F, P],
    "mogan_softmax_bwd": [P, P, P, P, L, I, L, F, P],
    "mogan_stn_fwd": [P, P, P, I, I, I, I, I, I, I, P],
    "mogan_stn_bwd": [P, P, P, I, I, I, I, I, I, I, P],
    "mogan_stn_fwd_ex": [P, P, P, I, I, I, I, I, I, I, I, I, I, P],
    "mogan_stn_bwd_ex": [P, P, P, I, I, I, I, I, I, I, I, I, I, P],
    "mogan_concat_fwd": [P, P, P, P, P, P, I, P, I, I, P],
    "mogan_concat_bwd": [P, P, P, P, P, P, P, I, I, I, P],
    "mogan_bbox_to_theta": [P, P, P, I, P],
    "mogan_attn_fwd": [P, P, P, P, P, I, I, I, I, I, P],
    "mogan_attn_bwd": [P, P, P, P, P, P, I, I, I, I, P],
    "mogan_bce_fwd": [P, F, F, P, I, I, P],
    "mogan_bce_bwd": [P, F, F, P, P, I, P],
    "mogan_bce_logits_fwd": [P, F, F, P, I, I, P],
    "mogan_bce_logits_bwd": [P, F, F, P, P, I, P],
    "mogan_kl_fwd": [P, P, P, I, P],
    "mogan_kl_bwd": [P, P, P, P, P, I, P],
    "mogan_reparam_fwd": [P, P, P, P, I, P],
    "mogan_reparam_bwd": [P, P, P, P, P, I, P],
    "mogan_maxpool_fwd": [P, P, P, I, I, I, I, I, P],
    "mogan_maxpool_bwd": [P, P, P, I, I, I, I, I, P],
    "mogan_avgpool_fwd": [P, P, I, I, I, I, I, I, P],
    "mogan_avgpool_bwd": [P, P, I, I, I, I, I, I, P],
    "mogan_bilinear_fwd": [P, P, I, I, I, I, I, P],
    "mogan_bilinear_bwd": [P, P, I, I, I, I, I, P],
    "mogan_adam_step": [P, P, P, P, P, L, F, F, F, F, I, P, I, F, F, P],
    "mogan_conv2d_affine_fwd_ex": [P, L, P, P, P, P, L, P, L, I] + [I] * 11 + [P, Z, P],
    "mogan_conv2d_fwd_ex": [P, L, P, P, L, P, L, I] + [I] * 10 + [P, Z, P],
    "mogan_conv2d_dgrad_ex": [P, L, P, P, L, P, L, I] + [I] * 10 + [P, Z, P],
    "mogan_conv2d_affine_fwd_group": [I, P, P, Z, P],
    "mogan_conv2d_dgrad_group": [I, P, P, Z, P],
    "mogan_pk_group": [I, P, P],
    "mogan_panel_tail_group": [I, P, P],
    "mogan_maxpool_fwd_ex": [P, P, L, P, I, I, I, I, I, I, P],
    "mogan_maxpool_bwd_ex": [P, P, L, P, P, I, I, I, I, I, I, I, P],
    "mogan_avgpool_bwd_ex": [P, P, P, I, I, I, I, I, I, I, P],
    "mogan_copy_strided": [P, L, P, L, I, L, P],
    "mogan_relu_bwd": [P, P, P, L, I, P],
    "mogan_feed_crop_flip": [P, P, P, P, I, I, I, P],
    "mogan_feed_resample": [P, P, P, P, P, I, I, I, I, P],
    "mogan_damsm_words_fwd": [P, P, P, I, I, I, I, I, F, F, F, P, P, P, P, P, P],
    "mogan_damsm_words_bwd": [P, P, P, P, P, P, P, I, I, I, I, I, F, F, F, P, P, P],
    "mogan_damsm_ce_fwd": [P, P, P, I, I, P, P, P, P, P],
    "mogan_damsm_ce_bwd": [P, P, P, P, P, I, I, P, P],
    "mogan_damsm_sent_fwd": [P, P, I, I, I, F, F, P, P],
    "mogan_damsm_sent_bwd": [P, P, P, I, I, I, F, F, P, P],
    "mogan_scalar_sum": [P, P, I, P, P],
    "mogan_scalar_scale": [P, P, I, P, P],
}


class ConvFwdArgs(ctypes.Structure):            # MoganConvFwdArgs (include/mogan_hip.h)
    _fields_ = [("x", P), ("x_bstride", L), ("w", P), ("scale", P), ("shift", P), ("y", P), ("y_bstride", L), ("y2", P),
                ("y2_bstride", L), ("msplit", I)] + [(k, I) for k in ("B", "Cin", "Hs", "Ws", "Cout", "KH", "KW", "stride",
                                                                      "ph", "pw", "relu")]


class ConvDgradArgs(ctypes.Structure):          # MoganConvDgradArgs
    _fields_ = [("dy", P), ("dy_bstride", L), ("w", P), ("dx", P), ("dx_bstride", L), ("relu_of", P), ("relu_bstride", L),
                ("accumulate", I)] + [(k, I) for k in ("B", "Cin", "Hs", "Ws", "Cout", "KH", "KW", "stride", "ph", "pw")]


class PkArgs(ctypes.Structure):                 # MoganPkArgs
    _fields_ = [("wpk", P), ("panel", P), ("raw", P)] + [(k, I) for k in ("B", "M", "Cp", "CGp", "cg0", "PH", "PW", "outH", "outW",
                                                                       "KH", "KW", "stride", "ph", "pw", "dgrad", "nsplit")]


class TailArgs(ctypes.Structure):               # MoganTailArgs
    _fields_ = [("src", P * 3), ("src_bs", L * 3), ("src_slab", L * 3), ("src_nsplit", I * 3), ("nsrc", I), ("add", P), ("add_bs", L),
                ("mask", P), ("mask_bs", L), ("scale", P), ("shift", P), ("relu", I), ("box", I), ("dst", P), ("dst_bs", L),
                ("panel", P), ("CGp", I), ("cg0", I), ("B", I), ("n", I), ("H", I), ("W", I)]


_RESTYPE = {"mogan_bn_ws_bytes": Z, "mogan_upconv3x3_ws_bytes": Z, "mogan_pk_weight_bytes": Z, "mogan_pk_panel_bytes": Z,
            "mogan_wino_prep_bytes": Z, "mogan_conv_prep_bytes": Z}
_ERRORS = {-1: "MOGAN_ERR_SHAPE", -2: "MOGAN_ERR_LAUNCH", -3: "MOGAN_ERR_WS"}

_lib = None
_ws = {}
WORKSPACE_BYTES = int(os.environ.get("MOGAN_WS_MB", "256")) << 20


class MoganHipError(RuntimeError):
    pass


def load():
    """Open libmogan_hip.so and type every entry point. Raises if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise MoganHipError(
                "libmogan_hip.so is not built (%s). Run `python __graft_entry__.py` / "
                "`python multiple-objects-gan_amd/build.py`; there is no CPU fallback." % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes = args
            fn.restype = _RESTYPE.get(name, I)
        _lib = lib
        _register_tuned(lib)
    return _lib


TUNED_CSV = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned_gemm_gfx950.csv")


def _register_tuned(lib):
    """Tuned (tile config, split-K) choices of the implicit-GEMM kernel for the GEMMs of the benchmark configurations
    (tools/tune_gemm.py, measured on MI355X).  MOGAN_TUNED=0 leaves the heuristic alone."""
    if os.environ.get("MOGAN_TUNED", "1") == "0" or not os.path.isfile(TUNED_CSV):
        return 0
    n = 0
    with open(TUNED_CSV) as f:
        for line in f:
            if line.startswith("#") or line.startswith("mode") or not line.strip():
                continue
            v = [int(x) for x in line.split(",")[:7]]
            if lib.mogan_gemm_tune_set(*v) == 0:
                n += 1
    return n


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


# ---------------------------------------------------------------------------------- hardware queues
# HIP multiplexes all streams of a process onto GPU_MAX_HW_QUEUES hardware queues (4 by default), assigned when a stream
# is first materialised; packets of streams that share a queue are processed in order, so one stream's event wait holds
# back its queue-mates.  The multi-stream train step (5 branch streams + weight-gradient side streams + RCCL's own) is
# very sensitive to which streams end up together: measured on the B=16 AttnGAN step, img/s by (queues, idle streams created
# first): (4,0) 299, (4,1) 290, (4,2) 305, (4,3) 277, (4,>=4) 288; (5,0) 219, (5,3) 306, (5,>=4) 185; (6..16, any) 156-280;
# (3,0) 291, (3,>=3) 300.5.  After torch.distributed's RCCL communicator is created (3 streams of its own) the default
# (4 queues) falls from 299 to 278.  (3 queues, >= 3 idle streams ahead of everything else) is the one plateau that does not
# move when more streams appear -- 300-301 img/s with and without the process group -- and is what the entry points use.
#
# Round 2, split-bf16 kernels (the MFMA kernels 25 % shorter, the arrangement matters more): one process, no process group:
# (4,0) 357-364 img/s in seven runs on three boxes, (4,2) 361, (3,3) 341-349, (4,1) 329, (4,3) 326-332, (4,4) 350, (5..8, 0)
# 257-264.  With the RCCL process group of a 1-GPU world: (3,3) 338-342, (4,3) 344-346, (4,0) 324-332.  A single process
# therefore runs on (4 queues, no idle streams) -- its stream creation order is fixed by this code alone --, a member of a
# process group kept the (3 queues, 3 idle streams) plateau.
#
# Round 3, discriminator branches as hipGraphs also under data parallelism (trainer.py, MOGAN_BRANCH_GRAPHS_DP): member of a
# 1-rank RCCL group (tools/dp_bg_probe.sh): (4,3) 395 / 395, (4,2) 395, (4,0) 366, (3,3) 364 img/s; eager branches (4,3) 373,
# (3,3) 370 -- a member of a process group now runs on (4 queues, 3 idle streams); RCCL's stream count on a real multi-GPU
# node is still unmeasured by the builder (GPU_MAX_HW_QUEUES / MOGAN_RESERVED_STREAMS override both).  One process without a
# group, same round (tools/queue_probe_single.sh, one box, interleaved): (4,2) 390.7 / 391.0, (4,0) 385.6 / 387.4, (4,3) 366, (4,1)
# 367 -- with the branch graphs the weight-gradient streams of the discriminators are idle and the best slot assignment moved;
# both kinds of process now use (4 queues, 2 idle streams).
_reserved = []


def hw_queue_defaults():
    """(GPU_MAX_HW_QUEUES, idle streams reserved first) of the eager multi-stream step: (4, 0) for every kind of process since
    round 5.  Rounds 1-4 bound the step's streams to hardware queues lazily (at their first use inside the step), so the layout
    depended on what else had created streams before -- RCCL's communicator in particular -- and the best setting differed by
    process kind (round 4: (4, 0) single process, (4, 3) member of a process group; +-8 %).  Now every entry point creates and
    touches the engine's streams in ONE fixed order right after torch.cuda.set_device, before torch.distributed exists
    (attngan/trainer.create_engine_streams, ENGINE_STREAM_ORDER): 437 img/s single process, 435 as member of a 1-rank RCCL group
    (profiles/r05_queue_table.csv).  MOGAN_HW_QUEUES (or GPU_MAX_HW_QUEUES itself) still overrides the queue count."""
    return (os.environ.get("MOGAN_HW_QUEUES", "4"), 0)


def configure_hw_queues():
    """Call BEFORE the first HIP call of the process (torch.cuda.set_device, library load): a GPU_MAX_HW_QUEUES set by the
    user wins."""
    os.environ.setdefault("GPU_MAX_HW_QUEUES", hw_queue_defaults()[0])


def reserve_hw_queues(n=None):
    """Call right after torch.cuda.set_device and before any other stream exists: n idle non-blocking HIP streams that
    take the first round of queue slots (0, the default of every kind of process since round 5, disables)."""
    if n is None:
        n = int(hw_queue_defaults()[1])
    if _reserved or n <= 0:
        return
    # through libmogan_hip.so, i.e. in the HIP runtime instance torch itself uses (a second copy of libamdhip64 loaded by
    # name would be another runtime with its own queues)
    got = load().mogan_reserve_streams(int(n))
    if got < n:
        raise RuntimeError("mogan_reserve_streams(%d) -> %d" % (n, got))
    _reserved.append(got)


def stream_ptr():
    """hipStream_t of the current torch stream.  torch.cuda.current_stream() costs ~7 us of python per call (3000 calls
    per train step); the raw-handle query is a single C call."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


_capturing = getattr(torch._C, "_cuda_isCurrentStreamCapturing", None) or torch.cuda.is_current_stream_capturing
_cap_epoch = [0, False]       # [number of capture sessions seen, was the previous workspace() call inside a capture]


def workspace(device):
    """Persistent split-K / reduction scratch per (device, stream).  Launches recorded into a hipGraph get a scratch buffer
    of their own, per stream and capture session: a replayed graph runs on whatever stream replays it, concurrently with
    eager work, and torch records every graph on ONE process-wide capture stream whose handle comes from the same
    32-stream pool as the engines' streams -- keyed by the handle alone, an eager stream with that handle and the replayed
    graph shared one buffer (seen as garbage image gradients from the graphed Inception branch once a long-lived test
    process had cycled through the pool).  Capture-session buffers are never handed to another session and stay allocated
    for the life of the process (they must outlive their graphs): MOGAN_WS_MB each, a handful of captures per process."""
    if device.type != "cuda":
        raise MoganHipError("mogan_hip ops need tensors on the GPU (got device %s): there is no CPU "
                            "path in the product" % device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    handle = _raw_stream(idx) if _raw_stream is not None else torch.cuda.current_stream(device).cuda_stream
    if _capturing():
        if not _cap_epoch[1]:
            _cap_epoch[0] += 1
            _cap_epoch[1] = True
        key = (idx, handle, _cap_epoch[0])
    else:
        _cap_epoch[1] = False
        key = (idx, handle)
    buf = _ws.get(key)
    if buf is None:
        buf = torch.empty(WORKSPACE_BYTES, dtype=torch.uint8, device=device)
        _ws[key] = buf
    return buf.data_ptr(), buf.numel()


def ptr(t):
    """Device pointer of a dense fp32/uint8/int32 CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise MoganHipError("mogan_hip ops need tensors on the GPU (got a %s tensor): "
                            "there is no CPU path in the product" % t.device)
    return t.data_ptr()


def call(name, *args):
    rc = getattr(load(), name)(*args)
    if rc != 0:
        raise MoganHipError("%s failed: %s" % (name, _ERRORS.get(rc, rc)))


def bn_ws_bytes(B, C, HW):
    return int(load().mogan_bn_ws_bytes(B, C, HW))


@contextlib.contextmanager
def capture_guard():
    """Wrap every hipGraph capture (torch.cuda.graph / make_graphed_callables) in this.  Python's cyclic GC may fire at any
    allocation; if it then finalises dead CUDA objects of an engine dropped earlier (graphs, events, a private memory pool)
    their hipFree / destroy calls land inside the capture, which HIP rejects -- raised from a destructor that is std::terminate
    (seen as "Fatal Python error: Aborted ... Garbage-collecting" in 2 of 5 full test-suite runs, never in a fresh process).
    Collect first, keep the collector off while capturing."""
    import gc
    gc.collect()
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()
