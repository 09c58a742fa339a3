"""ctypes binding of librsbwt.so (declared in include/rsbwt.h)."""
import ctypes as C
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.environ.get("RSBWT_LIB") or os.path.join(_HERE, "lib", "librsbwt.so")
_lock = threading.Lock()
_lib = None

RSBWT_OK = 0
RSBWT_ENODEV = -5


class RsbwtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rsbwt error {code}: {msg}")
        self.code = code


def lib_path():
    return _LIB


def build(force=False):
    """Compile csrc/ into lib/librsbwt.so for gfx950 (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(_HERE, "csrc", f) for f in os.listdir(os.path.join(_HERE, "csrc"))]
    srcs.append(os.path.join(_HERE, "..", "include", "rsbwt.h"))
    if not force and os.path.exists(_LIB):
        newest = max(os.path.getmtime(s) for s in srcs)
        if os.path.getmtime(_LIB) >= newest:
            return _LIB
    subprocess.check_call(["bash", os.path.join(_HERE, "build.sh")])
    return _LIB


_u64p = C.POINTER(C.c_uint64)
_vp = C.c_void_p

# name -> (restype, argtypes); must list every function of include/rsbwt.h
SIGNATURES = {
    "rsbwt_version": (C.c_char_p, []),
    "rsbwt_device_count": (C.c_int, []),
    "rsbwt_last_error": (C.c_char_p, []),
    "rsbwt_strerror": (C.c_char_p, [C.c_int]),
    "rsbwt_open": (C.c_int, [C.c_char_p, C.c_int, C.c_uint32, C.POINTER(_vp)]),
    "rsbwt_open_runs": (C.c_int, [_vp, C.c_uint64, C.c_uint64, C.c_int, C.c_uint32, C.POINTER(_vp)]),
    "rsbwt_open_device_runs": (C.c_int, [_vp, C.c_uint64, C.c_uint64, C.c_int, C.c_uint32, C.POINTER(_vp)]),
    "rsbwt_close": (None, [_vp]),
    "rsbwt_bwlen": (C.c_uint64, [_vp]),
    "rsbwt_pc": (C.c_uint64, [_vp, C.c_char]),
    "rsbwt_f": (C.c_char, [_vp, C.c_uint64]),
    "rsbwt_occ": (C.c_int, [_vp, C.c_char, C.c_uint64, _u64p]),
    "rsbwt_char": (C.c_int, [_vp, C.c_uint64, C.c_char_p]),
    "rsbwt_occ_at": (C.c_int, [_vp, C.c_char, C.c_uint64, _u64p]),
    "rsbwt_occ_batch": (C.c_int, [_vp, _vp, _vp, C.c_size_t, _vp]),
    "rsbwt_char_batch": (C.c_int, [_vp, _vp, C.c_size_t, _vp]),
    "rsbwt_occ_at_batch": (C.c_int, [_vp, _vp, _vp, C.c_size_t, _vp]),
    "rsbwt_num_runs": (C.c_uint64, [_vp]),
    "rsbwt_num_strings": (C.c_uint64, [_vp]),
    "rsbwt_num_lines": (C.c_uint64, [_vp]),
    "rsbwt_ktab_depth": (C.c_uint32, [_vp]),
    "rsbwt_window_span": (C.c_uint32, [_vp]),
    "rsbwt_far_lines": (C.c_uint64, [_vp]),
    "rsbwt_spilled_symbols": (C.c_uint64, [_vp]),
    "rsbwt_hbm_bytes": (C.c_uint64, [_vp]),
    "rsbwt_psi_hint_lines": (C.c_uint64, [_vp]),
    "rsbwt_opened_for_reads": (C.c_int, [_vp]),
    "rsbwt_prepare_extraction": (C.c_int, [_vp]),
    "rsbwt_debug_peek": (C.c_int, [_vp, C.c_int, C.c_uint64, _vp, C.c_size_t]),
    "rsbwt_attach_ktab": (C.c_int, [_vp, C.c_uint32]),
    "rsbwt_attach_ktab_format": (C.c_int, [_vp, C.c_uint32, C.c_uint32]),
    "rsbwt_ktab_info": (C.c_int, [_vp, C.POINTER(C.c_uint32), _u64p, _u64p]),
    "rsbwt_device": (C.c_int, [_vp]),
    "rsbwt_logical_device": (C.c_int, [_vp]),
    "rsbwt_find_intervals": (C.c_int, [_vp, _vp, C.c_size_t, C.c_uint32, C.c_size_t, _vp, _vp]),
    "rsbwt_count": (C.c_int, [_vp, _vp, C.c_size_t, C.c_uint32, C.c_size_t, _vp]),
    "rsbwt_find_intervals_1mm": (C.c_int, [_vp, _vp, C.c_size_t, C.c_uint32, C.c_size_t, _vp, _vp]),
    "rsbwt_hits_1mm": (C.c_int, [_vp, _vp, C.c_size_t, C.c_uint32, C.c_size_t, _vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "rsbwt_extract": (C.c_int, [_vp, _vp, C.c_size_t, _vp, C.c_uint32, _vp, _vp]),
    "rsbwt_query_exactmatch": (C.c_int, [_vp, _vp, C.c_size_t, C.c_uint32, C.c_size_t, _vp]),
    "rsbwt_query": (C.c_int, [_vp, _vp, C.c_size_t, C.c_uint32, C.c_size_t, _vp, _vp, C.c_uint32, _vp, C.c_size_t,
                              C.POINTER(C.c_size_t)]),
    "rsbwt_pack_kmers_dev": (C.c_int, [_vp, C.c_size_t, C.c_uint32, C.c_size_t, _vp, _vp, C.c_int, _vp]),
    "rsbwt_find_intervals_dev": (C.c_int, [_vp, _vp, _vp, C.c_size_t, C.c_uint32, _vp, _vp, _vp]),
    "rsbwt_count_dev": (C.c_int, [_vp, _vp, _vp, C.c_size_t, C.c_uint32, _vp, _vp]),
    "rsbwt_find_interval_pairs_dev": (C.c_int, [_vp, _vp, _vp, C.c_size_t, C.c_uint32, _vp, _vp]),
    "rsbwt_1mm_scratch_bytes": (C.c_size_t, [_vp, C.c_size_t, C.c_uint32]),
    "rsbwt_find_intervals_1mm_dev": (C.c_int, [_vp, _vp, _vp, C.c_size_t, C.c_uint32, _vp, _vp, _vp, _vp]),
    "rsbwt_packed_pairs_bytes": (C.c_size_t, [C.c_size_t]),
    "rsbwt_pack_interval_pairs_dev": (C.c_int, [_vp, C.c_size_t, _vp, _vp, C.c_int, _vp]),
    "rsbwt_unpack_interval_pairs_dev": (C.c_int, [_vp, C.c_size_t, _vp, C.c_int, _vp]),
    "rsbwt_pack_reads_dev": (C.c_int, [_vp, _vp, C.c_size_t, C.c_uint32, _vp, C.c_int, _vp]),
    "rsbwt_unpack_reads_dev": (C.c_int, [_vp, _vp, C.c_size_t, C.c_uint32, _vp, C.c_int, _vp]),
    "rsbwt_hits_1mm_scratch_bytes": (C.c_size_t, [_vp, C.c_size_t, C.c_uint32]),
    "rsbwt_hits_1mm_dev": (C.c_int, [_vp, _vp, _vp, C.c_size_t, C.c_uint32, _vp, C.c_size_t, _vp, _vp, _vp]),
    "rsbwt_set_1mm_scratch_bytes": (C.c_size_t, [_vp, C.c_size_t, C.c_uint32]),
    "rsbwt_set_find_intervals_1mm_dev": (C.c_int, [_vp, _vp, _vp, C.c_size_t, C.c_uint32, _vp, _vp, _vp, _vp]),
    "rsbwt_extract_dev": (C.c_int, [_vp, _vp, C.c_size_t, _vp, C.c_uint32, _vp, _vp, _vp]),
    "rsbwt_last_search_ms": (C.c_int, [_vp, C.POINTER(C.c_float)]),
    "rsbwt_search_history_ms": (C.c_int, [_vp, C.POINTER(C.c_float), C.c_size_t, C.POINTER(C.c_size_t)]),
    "rsbwt_set_counting": (C.c_int, [_vp, C.c_int]),
    "rsbwt_last_search_work": (C.c_int, [_vp, _u64p, _u64p, _u64p]),
    "rsbwt_last_search_counters": (C.c_int, [_vp, _u64p]),
    "rsbwt_last_search_phases": (C.c_int, [_vp, _u64p, _u64p]),
    "rsbwt_last_search_ktab_lookups": (C.c_int, [_vp, _u64p]),
    "rsbwt_synth_runs_dev": (C.c_int, [_vp, C.c_uint64, C.c_uint64, C.c_int, _vp]),
    "rsbwt_synth_runs_dev_at": (C.c_int, [_vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, _vp]),
    "rsbwt_synth_runs_host": (C.c_int, [_vp, C.c_uint64, C.c_uint64]),
    "rsbwt_sample_present_kmers_dev": (C.c_int, [_vp, C.c_size_t, C.c_uint32, C.c_size_t, C.c_uint64, _vp, _vp]),
    "rsbwt_bpi2_write": (C.c_int, [C.c_char_p, C.c_char_p]),
    "rsbwt_bpi2_check": (C.c_int, [C.c_void_p, C.c_char_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "rsbwt_bpi2_validate_file": (C.c_int, [C.c_char_p]),
    "rsbwt_synth_popbwt": (C.c_int, [C.c_char_p, C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_double,
                                     C.c_uint32, C.c_double, C.c_int, C.c_int]),
    "rsbwt_service_counts": (C.c_int, [_vp, _vp, C.c_size_t, _vp, C.c_size_t, _vp, C.c_size_t, _vp, C.POINTER(C.c_size_t)]),
    "rsbwt_service_config_load": (C.c_int, [C.c_char_p, C.POINTER(_vp)]),
    "rsbwt_service_config_free": (None, [_vp]),
    "rsbwt_service_config_get": (C.c_char_p, [_vp, C.c_char_p]),
    "rsbwt_service_config_array_len": (C.c_size_t, [_vp, C.c_char_p]),
    "rsbwt_service_config_array_item": (C.c_char_p, [_vp, C.c_char_p, C.c_size_t]),
    "rsbwt_transport_inproc": (C.c_int, [C.POINTER(_vp)]),
    "rsbwt_transport_zmq": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(_vp)]),
    "rsbwt_zmq_available": (C.c_int, []),
    "rsbwt_transport_free": (None, [_vp]),
    "rsbwt_transport_push_request": (C.c_int, [_vp, _vp, C.c_size_t]),
    "rsbwt_transport_pop_reply": (C.c_int, [_vp, C.c_int, _vp, C.c_size_t, C.POINTER(C.c_size_t), C.c_int64]),
    "rsbwt_transport_close": (None, [_vp]),
    "rsbwt_service_create": (C.c_int, [_vp, _vp, C.c_int64, C.c_size_t, C.c_int, C.POINTER(_vp)]),
    "rsbwt_service_set_workers": (None, [_vp, C.c_int]),
    "rsbwt_transport_push_requests": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "rsbwt_transport_pop_replies": (C.c_int, [_vp, C.c_int, _vp, C.c_size_t, _vp, C.c_size_t, C.POINTER(C.c_size_t), C.c_int64]),
    "rsbwt_service_set_other_handler": (None, [_vp, _vp, _vp]),
    "rsbwt_service_set_reads": (None, [_vp, C.c_int, C.c_uint32, C.c_uint32]),
    "rsbwt_service_set_suffixes": (C.c_int, [_vp, C.POINTER(C.c_char_p), C.c_size_t]),
    "rsbwt_service_read_requests": (C.c_uint64, [_vp]),
    "rsbwt_service_run": (C.c_int, [_vp]),
    "rsbwt_service_start": (C.c_int, [_vp]),
    "rsbwt_service_stop": (C.c_int, [_vp]),
    "rsbwt_service_free": (None, [_vp]),
    "rsbwt_service_stats": (None, [_vp, _u64p]),
    "rsbwt_proto_decode_request": (C.c_int, [_vp, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                             C.POINTER(C.c_char_p), C.POINTER(C.c_size_t)]),
    "rsbwt_proto_encode_count_reply": (C.c_size_t, [_vp, C.c_size_t, C.c_int, C.c_char_p, C.c_size_t, C.c_int, C.c_int32]),
    "rsbwt_proto_encode_reads_reply": (C.c_size_t, [_vp, C.c_size_t, C.c_int, C.c_char_p, C.c_size_t, C.c_int,
                                                    C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_size_t]),
    "rsbwt_set_open": (C.c_int, [C.POINTER(C.c_char_p), C.c_size_t, C.POINTER(C.c_int), C.c_uint32, C.POINTER(_vp)]),
    "rsbwt_set_from_handles": (C.c_int, [C.POINTER(_vp), C.c_size_t, C.POINTER(_vp)]),
    "rsbwt_set_close": (None, [_vp]),
    "rsbwt_set_size": (C.c_size_t, [_vp]),
    "rsbwt_set_devices": (C.c_size_t, [_vp]),
    "rsbwt_set_attach_ktabs": (C.c_int, [_vp, C.c_uint32]),
    "rsbwt_set_attach_ktabs_format": (C.c_int, [_vp, C.c_uint32, C.c_uint32]),
    "rsbwt_set_auto_ktab_depth": (C.c_uint32, [_vp]),
    "rsbwt_set_auto_ktab": (C.c_int, [_vp, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "rsbwt_auto_ktab_for_budget": (C.c_int, [C.c_uint64, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "rsbwt_set_find_intervals_dev": (C.c_int, [_vp, _vp, _vp, C.c_size_t, C.c_uint32, _vp, _vp, _vp]),
    "rsbwt_set_count_dev": (C.c_int, [_vp, _vp, _vp, C.c_size_t, C.c_uint32, _vp, _vp]),
    "rsbwt_set_find_interval_pairs_dev": (C.c_int, [_vp, _vp, _vp, C.c_size_t, C.c_uint32, _vp, _vp]),
    "rsbwt_set_gather_intervals_dev": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(C.c_size_t), _vp, C.POINTER(_vp)]),
    "rsbwt_set_hits_1mm": (C.c_int, [_vp, _vp, C.c_size_t, C.c_uint32, C.c_size_t, _vp, C.c_size_t, _vp, C.POINTER(C.c_size_t)]),
    "rsbwt_set_extract": (C.c_int, [_vp, _vp, _vp, C.c_size_t, _vp, C.c_uint32, _vp, _vp]),
    "rsbwt_set_query": (C.c_int, [_vp, _vp, C.c_size_t, C.c_uint32, C.c_size_t, _vp, _vp, _vp, C.c_uint32, _vp, C.c_size_t,
                                  C.POINTER(C.c_size_t)]),
    "rsbwt_set_find_intervals_var": (C.c_int, [_vp, _vp, _vp, C.c_size_t, _vp, _vp]),
    "rsbwt_set_count_var": (C.c_int, [_vp, _vp, _vp, C.c_size_t, _vp]),
    "rsbwt_set_query_var": (C.c_int, [_vp, _vp, _vp, C.c_size_t, _vp, _vp, _vp, C.c_uint32, _vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "rsbwt_set_hits_1mm_scratch_bytes": (C.c_size_t, [_vp, C.c_size_t, C.c_uint32]),
    "rsbwt_set_hits_1mm_is_fused": (C.c_int, [_vp, C.c_size_t, C.c_uint32]),
    "rsbwt_set_hits_1mm_dev": (C.c_int, [_vp, _vp, _vp, C.c_size_t, C.c_uint32, _vp, C.c_size_t, _vp, _vp, _vp]),
    "rsbwt_set_extract_dev": (C.c_int, [_vp, _vp, C.c_size_t, _vp, C.c_uint32, _vp, _vp, _vp]),
    "rsbwt_set_records_bytes": (C.c_size_t, [_vp, C.c_size_t]),
    "rsbwt_set_prepare_dev": (C.c_int, [_vp, _vp, _vp, C.c_size_t, C.c_uint32, _vp, _vp]),
    "rsbwt_set_find_interval_pairs_prepared_dev": (C.c_int, [_vp, _vp, _vp, _vp, C.c_size_t, C.c_uint32, _vp, _vp]),
    "rsbwt_rccl_available": (C.c_int, []),
    "rsbwt_set_set_counting": (C.c_int, [_vp, C.c_int]),
    "rsbwt_set_search_history_ms": (C.c_int, [_vp, C.POINTER(C.c_float), C.c_size_t, C.POINTER(C.c_size_t)]),
    "rsbwt_set_last_search_counters": (C.c_int, [_vp, _u64p]),
    "rsbwt_layout_selftest_host": (C.c_int, [_vp, C.c_uint64, C.c_uint32, _u64p, _u64p]),
    "rsbwt_layout_selftest_psi_host": (C.c_int, [_vp, C.c_uint64, C.c_uint32, _u64p, _u64p]),
    "rsbwt_ktab_group_selftest_host": (C.c_int, [_vp, _vp, C.c_size_t, _vp]),
    "rsbwt_debug_fast_window": (C.c_int, [_vp, C.c_size_t, C.c_uint32, _vp, _vp, C.c_int]),
    "rsbwt_debug_poke": (C.c_int, [_vp, C.c_int, C.c_uint64, _vp, C.c_size_t]),
    "rsbwt_set_shard": (_vp, [_vp, C.c_size_t]),
    "rsbwt_set_find_intervals": (C.c_int, [_vp, _vp, C.c_size_t, C.c_uint32, C.c_size_t, _vp, _vp]),
    "rsbwt_set_count": (C.c_int, [_vp, _vp, C.c_size_t, C.c_uint32, C.c_size_t, _vp]),
}


def lib():
    """The loaded library.  Fails loudly when it has not been built: no fallback exists."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(_LIB):
                raise RuntimeError(
                    f"{_LIB} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                    "(or readserver_amd/build.sh). The popBWT engine is the HIP library only; "
                    "there is no CPU fallback.")
            # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64.so.7.
            # If torch is installed, load it first so that librsbwt's libamdhip64.so.7 dependency
            # binds to the copy torch uses (two runtimes in one process cannot both own the GPU).
            if os.environ.get("RSBWT_NO_TORCH_PRELOAD") != "1":
                try:
                    import torch  # noqa: F401
                except ImportError:
                    pass
            L = C.CDLL(_LIB)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(L, name)
                fn.restype = res
                fn.argtypes = args
            _lib = L
        return _lib


def check(rc):
    if rc != RSBWT_OK:
        msg = lib().rsbwt_last_error().decode(errors="replace")
        raise RsbwtError(rc, msg or lib().rsbwt_strerror(rc).decode())
