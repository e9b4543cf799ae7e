"""ctypes binding of libkpop_hip.so -- the C ABI declared in include/kpop_hip.h.

There is no CPU fallback: if the shared library is missing, or no GPU is
visible when kpop_init() runs, the caller gets an exception.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KPOP_HIP_LIB", os.path.join(_HERE, "libkpop_hip.so"))  # the override is for A/B runs of two builds

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
f64p = C.POINTER(C.c_double)
vp = C.c_void_p


class PipelineConfig(C.Structure):
    """kpop_pipeline_config (include/kpop_hip.h)"""
    _fields_ = [("struct_size", C.c_uint32), ("content", C.c_int), ("normalize_counts", C.c_int), ("kind", C.c_int),
                ("p", C.c_double), ("normalize_distances", C.c_int), ("outputs", C.c_int), ("keep_at_most", C.c_uint32),
                ("max_neighbours", C.c_uint32), ("chunk_reads", C.c_uint32), ("depth", C.c_uint32),
                ("chunk_bases", C.c_uint64), ("record_timeline", C.c_int)]


class PipelineOutputs(C.Structure):
    """kpop_pipeline_outputs (include/kpop_hip.h)"""
    _fields_ = [("twisted", vp), ("distances", vp), ("stats", vp), ("n_neighbours", vp), ("nb_index", vp),
                ("nb_distance", vp), ("nb_z", vp)]


# name -> (restype, argtypes); mirrors include/kpop_hip.h one to one
SIGNATURES = {
    "kpop_init": (C.c_int, [C.c_int]),
    "kpop_init_devices": (C.c_int, [C.POINTER(C.c_int), C.c_int]),
    "kpop_use_device": (C.c_int, [C.c_int]),
    "kpop_device_slots": (C.c_int, []),
    "kpop_host_alloc": (C.c_int, [C.POINTER(vp), C.c_uint64]),
    "kpop_host_free": (C.c_int, [vp]),
    "kpop_host_register": (C.c_int, [vp, C.c_uint64]),
    "kpop_host_unregister": (C.c_int, [vp]),
    "kpop_pipeline_create": (C.c_int, [vp, f64p, C.c_uint32, f64p, C.POINTER(PipelineConfig), C.POINTER(vp)]),
    "kpop_pipeline_submit": (C.c_int, [vp, vp, vp, C.c_uint32, C.POINTER(PipelineOutputs), u64p]),
    "kpop_pipeline_submit_packed": (C.c_int, [vp, vp, vp, vp, C.c_uint32, C.POINTER(PipelineOutputs), u64p]),
    "kpop_pipeline_collect": (C.c_int, [vp, C.c_uint64]),
    "kpop_packed_code_words": (C.c_uint64, [C.c_uint64]),
    "kpop_packed_mask_words": (C.c_uint64, [C.c_uint64]),
    "kpop_pack_bases": (C.c_int, [vp, C.c_uint64, vp, vp, C.c_int]),
    "kpop_count_twist_packed": (C.c_int, [vp, vp, vp, u64p, C.c_uint32, C.c_int, C.c_int, f64p]),
    "kpop_dev_unpack_bases": (C.c_int, [vp, vp, C.c_uint64, vp, vp]),
    "kpop_dev_count_twist_packed": (C.c_int, [vp, vp, vp, vp, C.c_uint32, C.c_uint64, C.c_uint32, C.c_int, C.c_int, vp, vp]),
    "kpop_pipeline_run": (C.c_int, [vp, vp, vp, C.c_uint32, C.POINTER(PipelineOutputs)]),
    "kpop_pipeline_stats": (C.c_int, [vp, u32p, C.POINTER(C.c_int), u32p]),
    "kpop_pipeline_timeline": (C.c_int, [vp, C.c_uint32, f64p, u32p]),
    "kpop_pipeline_destroy": (C.c_int, [vp]),
    "kpop_dev_workspace_reserve_stream": (C.c_int, [C.c_uint64, vp]),
    "kpop_shard_bounds": (C.c_int, [C.c_uint64, C.c_int, C.c_int, u64p, u64p]),
    "kpop_twister_replicate": (C.c_int, [vp, C.c_int, C.POINTER(vp)]),
    "kpop_sharded_create": (C.c_int, [vp, f64p, C.c_uint32, f64p, C.POINTER(PipelineConfig), C.POINTER(vp)]),
    "kpop_sharded_slots": (C.c_int, [vp]),
    "kpop_sharded_run": (C.c_int, [vp, vp, vp, C.c_uint32, C.POINTER(PipelineOutputs)]),
    "kpop_sharded_spectra_twist": (C.c_int, [vp, vp, vp, C.c_uint32, C.c_int, C.c_int, C.c_int, f64p]),
    "kpop_sharded_resident_step": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp), u32p, u64p, C.c_uint32, C.c_int, C.c_int]),
    "kpop_sharded_resident_buffers": (C.c_int, [vp, C.c_int, C.POINTER(vp), u64p, u64p, C.POINTER(vp)]),
    "kpop_sharded_timings": (C.c_int, [vp, C.c_int, f64p, f64p]),
    "kpop_sharded_chunk_timings": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int)]),
    "kpop_sharded_all_vs_all_summary": (C.c_int, [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, u64p, u64p, f64p, u32p,
                                                  u32p, f64p, f64p]),
    "kpop_sharded_destroy": (C.c_int, [vp]),
    "kpop_sharded_distance_rowwise": (C.c_int, [f64p, C.c_uint32, f64p, C.c_uint32, C.c_uint32, f64p, C.c_int, C.c_double,
                                                C.c_int, f64p]),
    "kpop_sharded_distance_summary": (C.c_int, [f64p, C.c_uint32, f64p, C.c_uint32, C.c_uint32, f64p, C.c_int, C.c_double,
                                                C.c_int, C.c_uint32, C.c_uint32, f64p, u32p, u32p, f64p, f64p]),
    "kpop_shutdown": (C.c_int, []),
    "kpop_device_count": (C.c_int, []),
    "kpop_last_error": (C.c_char_p, []),
    "kpop_version": (C.c_char_p, []),
    "kpop_synchronize": (C.c_int, [vp]),
    "kpop_tune": (C.c_int, [C.c_char_p, C.c_int]),
    "kpop_debug_counters": (C.c_int, [C.POINTER(C.c_uint64), C.c_int]),
    "kpop_debug_summary_fallbacks": (C.c_int, [C.POINTER(C.c_uint64)]),
    "kpop_dev_malloc": (C.c_int, [C.POINTER(vp), C.c_uint64]),
    "kpop_dev_free": (C.c_int, [vp]),
    "kpop_memcpy_h2d": (C.c_int, [vp, vp, C.c_uint64]),
    "kpop_memcpy_d2h": (C.c_int, [vp, vp, C.c_uint64]),
    "kpop_dev_memset": (C.c_int, [vp, C.c_int, C.c_uint64]),
    "kpop_count_reads": (C.c_int, [u8p, u64p, C.c_uint32, C.c_int, C.c_int, C.c_int, u64p, u32p, u64p,
                                   C.c_uint64]),
    "kpop_twister_load": (C.c_int, [f64p, C.c_uint64, C.c_uint32, u64p, C.c_int, C.POINTER(vp)]),
    "kpop_twister_synth": (C.c_int, [C.c_uint64, C.c_int, C.c_int, C.c_uint32, C.POINTER(vp)]),
    "kpop_twister_synth_slice": (C.c_int, [C.c_uint64, C.c_int, C.c_int, C.c_uint32, C.c_uint64, C.c_uint64, C.c_int,
                                           C.POINTER(vp)]),
    "kpop_twister_set_count_k": (C.c_int, [vp, C.c_int]),
    "kpop_twister_free": (C.c_int, [vp]),
    "kpop_twister_info": (C.c_int, [vp, u64p, u32p, C.POINTER(C.c_int), u64p]),
    "kpop_twister_direct_bytes": (C.c_int, [vp, u64p]),
    "kpop_twist": (C.c_int, [vp, u64p, f64p, u64p, C.c_uint32, C.c_int, f64p]),
    "kpop_count_twist": (C.c_int, [vp, u8p, u64p, C.c_uint32, C.c_int, C.c_int, f64p]),
    "kpop_spectra_twist": (C.c_int, [vp, u8p, u64p, C.c_uint32, C.c_int, C.c_int, C.c_int, f64p]),
    "kpop_ca": (C.c_int, [f64p, C.c_uint64, C.c_uint32, C.c_int, u32p, f64p, f64p, f64p]),
    "kpop_metric_compute": (C.c_int, [C.c_int, f64p, C.c_uint32, C.c_double, C.c_double, C.c_double, f64p]),
    "kpop_distance_rowwise": (C.c_int, [f64p, C.c_uint32, f64p, C.c_uint32, C.c_uint32, f64p, C.c_int,
                                        C.c_double, C.c_int, f64p]),
    "kpop_distance_summary": (C.c_int, [f64p, C.c_uint32, f64p, C.c_uint32, C.c_uint32, f64p, C.c_int,
                                        C.c_double, C.c_int, C.c_uint32, C.c_uint32, f64p, u32p, u32p, f64p,
                                        f64p]),
    "kpop_embeddings": (C.c_int, [f64p, C.c_uint32, C.c_uint32, f64p, C.c_int, C.c_double, C.c_int, f64p]),
    "kpop_splits_gaps": (C.c_int, [f64p, C.c_uint32, C.c_uint32, C.c_uint32, u32p, f64p, u32p, u32p, u32p]),
    "kpop_summarize_distances": (C.c_int, [f64p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, f64p, u32p, u32p, f64p,
                                           f64p]),
    "kpop_dev_summarize_distances": (C.c_int, [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp, vp, vp,
                                               vp]),
    "kpop_counter_stats": (C.c_int, [C.POINTER(vp), C.c_uint32, C.c_uint64, C.c_double, C.c_double, f64p, f64p]),
    "kpop_counter_combine": (C.c_int, [C.POINTER(vp), C.c_uint64, u32p, C.c_uint32, f64p, C.c_int, vp, f64p]),
    "kpop_counter_transform": (C.c_int, [C.POINTER(vp), C.c_uint32, C.c_uint64, C.c_int, C.c_double, C.c_double, f64p,
                                         C.c_int, f64p]),
    "kpop_dev_counter_ld": (C.c_uint64, [C.c_uint64]),
    "kpop_dev_counter_workspace_bytes": (C.c_uint64, [C.c_uint32, C.c_uint64]),
    "kpop_dev_counter_stats": (C.c_int, [vp, C.c_uint64, C.c_uint32, C.c_uint64, C.c_double, C.c_double, vp, vp, vp, vp]),
    "kpop_dev_counter_combine": (C.c_int, [vp, C.c_uint64, C.c_uint64, vp, vp, C.c_uint32, C.c_uint32, C.c_double,
                                           C.c_int, vp, vp, vp, vp]),
    "kpop_dev_counter_transform": (C.c_int, [vp, C.c_uint64, C.c_uint32, C.c_uint64, C.c_int, C.c_double, C.c_double,
                                             vp, C.c_int, vp, vp]),
    "kpop_dev_division_probe": (C.c_int, [vp, vp, C.c_uint64, vp, vp, vp]),
    "kpop_dev_synth_reads": (C.c_int, [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, vp, vp, vp]),
    "kpop_dev_count_reads_scratch_bytes": (C.c_uint64, [C.c_uint32, C.c_uint32, C.c_int]),
    "kpop_dev_count_reads": (C.c_int, [vp, vp, C.c_uint32, C.c_uint32, C.c_int, C.c_int, vp, vp, vp, vp, vp]),
    "kpop_dev_workspace_reserve": (C.c_int, [C.c_uint64]),
    "kpop_dev_count_twist": (C.c_int, [vp, vp, vp, C.c_uint32, C.c_uint64, C.c_uint32, C.c_int, C.c_int, vp, vp]),
    "kpop_dev_twist": (C.c_int, [vp, vp, vp, vp, C.c_uint32, C.c_uint64, C.c_int, vp, vp]),
    "kpop_dev_ca_workspace_bytes": (C.c_uint64, [C.c_uint64, C.c_uint32]),
    "kpop_dev_ca": (C.c_int, [vp, C.c_uint64, C.c_uint32, C.c_int, vp, u32p, vp, vp, vp, vp]),
    "kpop_dev_table_row_sums": (C.c_int, [vp, C.c_uint64, C.c_uint32, vp, vp]),
    "kpop_dev_table_col_sums": (C.c_int, [vp, C.c_uint64, C.c_uint32, vp, vp]),
    "kpop_dev_table_gather_rows": (C.c_int, [vp, C.c_uint32, vp, C.c_uint64, vp, vp]),
    "kpop_dev_twist_dense_workspace_bytes": (C.c_uint64, [vp, C.c_uint32]),
    "kpop_dev_twist_dense": (C.c_int, [vp, vp, vp, vp, C.c_uint32, C.c_int, vp, vp, vp]),
    "kpop_dev_count_twist_dense_workspace_bytes": (C.c_uint64, [vp, C.c_uint32]),
    "kpop_dev_count_twist_dense": (C.c_int, [vp, vp, vp, C.c_uint32, C.c_int, C.c_int, vp, vp, vp]),
    "kpop_dev_twist_dense_sorted": (C.c_int, [vp, vp, vp, vp, C.c_uint32, C.c_int, vp, vp, vp]),
    "kpop_dev_distance_workspace_bytes": (C.c_uint64, [C.c_uint32, C.c_uint32, C.c_uint32]),
    "kpop_dev_distance_rowwise": (C.c_int, [vp, C.c_uint32, vp, C.c_uint32, C.c_uint32, vp, C.c_int,
                                            C.c_double, C.c_int, vp, vp, vp]),
    "kpop_dev_row_norms": (C.c_int, [vp, C.c_uint32, C.c_uint32, vp, C.c_int, C.c_double, vp, vp]),
    "kpop_dev_distance_rowwise_norms": (C.c_int, [vp, C.c_uint32, vp, vp, C.c_uint32, C.c_uint32, vp, C.c_int, C.c_double, C.c_int, vp, vp,
                                                  vp]),
    "kpop_dev_distance_summary": (C.c_int, [vp, C.c_uint32, vp, C.c_uint32, C.c_uint32, vp, C.c_int,
                                            C.c_double, C.c_int, C.c_uint32, C.c_uint32, vp, vp, vp, vp, vp,
                                            vp, vp]),
}


class KPopError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libkpop_hip: %s (status %d)" % (msg, code))
        self.code = code


_lib = None


def load():
    """dlopen libkpop_hip.so and declare every symbol of include/kpop_hip.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C kpop_amd/csrc` (there is no CPU fallback)" % LIB_PATH)
    # PyTorch-ROCm wheels bundle their own HIP/HSA runtime.  Two runtimes in one process do not both see the
    # GPU, and whichever is loaded first wins, so when torch is installed let its copy be the process's runtime
    # (libkpop_hip.so only needs libamdhip64.so.7 by SONAME).  Stand-alone users (the C++ CLIs) get /opt/rocm's.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        if "KPOP_HIP_LIB" in os.environ and not hasattr(lib, name):
            continue  # an older build under A/B comparison
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise KPopError(rc, load().kpop_last_error().decode("utf-8", "replace"))
