"""ctypes binding of libsourmash_amd.so -- the same way Python sourmash binds the reference
through cffi (reference README.md:28-31): only the C ABI of include/sourmash.h (+ the additive
include/sourmash_amd.h) is used.  No fallback: a missing library is an ImportError, a missing
GPU surfaces as SourmashError(code 2) from the first call that needs the device."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("SOURMASH_AMD_LIB") or os.path.join(HERE, "lib", "libsourmash_amd.so")   # override: experiments

u64p = C.POINTER(C.c_uint64)
f64p = C.POINTER(C.c_double)


class SourmashStr(C.Structure):
    _fields_ = [("data", C.c_void_p), ("len", C.c_size_t), ("owned", C.c_bool)]


class SmhCompareTuning(C.Structure):
    _fields_ = [("route", C.c_uint32), ("visit_all_tiles", C.c_uint32), ("use_symmetry", C.c_uint32),
                ("comp_pairs_limit", C.c_uint64), ("split_frequent", C.c_uint32), ("dictionary", C.c_uint32),
                ("no_range_masks", C.c_uint32)]


class SmhCompareStats(C.Structure):
    _fields_ = [("route", C.c_uint32), ("rows_per_tile", C.c_uint32), ("tiles_visited", C.c_uint64),
                ("tiles_total", C.c_uint64), ("pairs_per_tile", C.c_uint64), ("lds_overflow_steps", C.c_uint64),
                ("frequent_hashes", C.c_uint32), ("pipelined", C.c_uint32),
                ("span_halvings", C.c_uint32), ("prefetched_after_halving", C.c_uint32)]


def build(force=False):
    """Compile the HIP/C++ sources in-tree (hipcc --offload-arch=gfx950)."""
    src = os.path.join(HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-C", src, "clean"])
    subprocess.check_call(["make", "-C", src, "-j8", "-s"])
    return SO_PATH


_SIGS = {
    # name: (restype, argtypes)
    "hash_murmur": (C.c_uint64, [C.c_char_p, C.c_uint64]),
    "kmerminhash_new": (C.c_void_p, [C.c_uint32, C.c_uint32, C.c_bool, C.c_uint64, C.c_uint64, C.c_bool]),
    "kmerminhash_free": (None, [C.c_void_p]),
    "kmerminhash_add_sequence": (None, [C.c_void_p, C.c_char_p, C.c_bool]),
    "kmerminhash_add_hash": (None, [C.c_void_p, C.c_uint64]),
    "kmerminhash_add_word": (None, [C.c_void_p, C.c_char_p]),
    "kmerminhash_add_from": (None, [C.c_void_p, C.c_void_p]),
    "kmerminhash_merge": (None, [C.c_void_p, C.c_void_p]),
    "kmerminhash_compare": (C.c_double, [C.c_void_p, C.c_void_p]),
    "kmerminhash_count_common": (C.c_uint64, [C.c_void_p, C.c_void_p]),
    "kmerminhash_intersection": (C.c_uint64, [C.c_void_p, C.c_void_p]),
    "kmerminhash_get_mins": (C.c_void_p, [C.c_void_p]),
    "kmerminhash_get_mins_size": (C.c_size_t, [C.c_void_p]),
    "kmerminhash_get_min_idx": (C.c_uint64, [C.c_void_p, C.c_uint64]),
    "kmerminhash_mins_push": (None, [C.c_void_p, C.c_uint64]),
    "kmerminhash_get_abunds": (C.c_void_p, [C.c_void_p]),
    "kmerminhash_get_abunds_size": (C.c_size_t, [C.c_void_p]),
    "kmerminhash_get_abund_idx": (C.c_uint64, [C.c_void_p, C.c_uint64]),
    "kmerminhash_abunds_push": (None, [C.c_void_p, C.c_uint64]),
    "kmerminhash_is_protein": (C.c_bool, [C.c_void_p]),
    "kmerminhash_seed": (C.c_uint64, [C.c_void_p]),
    "kmerminhash_track_abundance": (C.c_bool, [C.c_void_p]),
    "kmerminhash_num": (C.c_uint32, [C.c_void_p]),
    "kmerminhash_ksize": (C.c_uint32, [C.c_void_p]),
    "kmerminhash_max_hash": (C.c_uint64, [C.c_void_p]),
    "signature_new": (C.c_void_p, []),
    "signature_free": (None, [C.c_void_p]),
    "signature_set_name": (None, [C.c_void_p, C.c_char_p]),
    "signature_set_filename": (None, [C.c_void_p, C.c_char_p]),
    "signature_push_mh": (None, [C.c_void_p, C.c_void_p]),
    "signature_set_mh": (None, [C.c_void_p, C.c_void_p]),
    "signature_get_name": (SourmashStr, [C.c_void_p]),
    "signature_get_filename": (SourmashStr, [C.c_void_p]),
    "signature_get_license": (SourmashStr, [C.c_void_p]),
    "signature_first_mh": (C.c_void_p, [C.c_void_p]),
    "signature_eq": (C.c_bool, [C.c_void_p, C.c_void_p]),
    "signature_save_json": (SourmashStr, [C.c_void_p]),
    "signature_get_mhs": (C.POINTER(C.c_void_p), [C.c_void_p, C.POINTER(C.c_size_t)]),
    "signatures_save_buffer": (SourmashStr, [C.POINTER(C.c_void_p), C.c_size_t]),
    "signatures_load_path": (C.POINTER(C.c_void_p), [C.c_char_p, C.c_bool, C.c_size_t, C.c_char_p, C.POINTER(C.c_size_t)]),
    "signatures_load_buffer": (C.POINTER(C.c_void_p), [C.c_char_p, C.c_size_t, C.c_bool, C.c_size_t, C.c_char_p, C.POINTER(C.c_size_t)]),
    "sourmash_err_clear": (None, []),
    "sourmash_err_get_backtrace": (SourmashStr, []),
    "sourmash_err_get_last_code": (C.c_uint32, []),
    "sourmash_err_get_last_message": (SourmashStr, []),
    "sourmash_init": (None, []),
    "sourmash_str_free": (None, [C.POINTER(SourmashStr)]),
    "sourmash_str_from_cstr": (SourmashStr, [C.c_char_p]),
    # additive (include/sourmash_amd.h)
    "smh_device_available": (C.c_int, []),
    "smh_device_info": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "smh_add_sequence_len": (C.c_int, [C.c_void_p, C.c_char_p, C.c_uint64, C.c_bool]),
    "smh_add_sequences": (C.c_int, [C.c_void_p, C.c_char_p, u64p, C.c_uint32, C.c_bool]),
    "smh_add_sequences_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, u64p, C.c_uint32, C.c_bool, C.c_void_p]),
    "smh_add_sequences_grouped": (C.c_int, [C.POINTER(C.c_void_p), C.c_uint32, C.c_char_p, u64p, C.POINTER(C.c_uint32), C.c_uint32, C.c_bool]),
    "smh_add_sequences_grouped_dev": (C.c_int, [C.POINTER(C.c_void_p), C.c_uint32, C.c_void_p, C.c_uint64, u64p, C.POINTER(C.c_uint32), C.c_uint32, C.c_bool, C.c_void_p]),
    "smh_add_many": (C.c_int, [C.c_void_p, u64p, C.c_uint64]),
    "smh_add_many_with_abund": (C.c_int, [C.c_void_p, u64p, u64p, C.c_uint64]),
    "smh_check_compatible": (C.c_int, [C.c_void_p, C.c_void_p]),
    "smh_intersection": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(u64p), u64p, u64p]),
    "smh_sketch_export_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, u64p, C.c_void_p]),
    "smh_sketch_absorb_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, u64p, u64p, C.c_uint32, C.c_void_p]),
    "smh_hash_words": (C.c_int, [C.c_char_p, u64p, C.c_uint32, C.c_uint64, u64p]),
    "smh_compare_block": (C.c_int, [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_void_p), C.c_uint32, f64p, u64p, u64p, u64p, f64p]),
    "smh_compare_block_dev": (C.c_int, [C.c_void_p, u64p, C.c_uint32, C.c_void_p, u64p, C.c_uint32, C.c_uint32,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "smh_collection_begin": (C.c_void_p, [C.c_void_p, u64p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]),
    "smh_collection_share_bytes": (C.c_uint64, [C.c_void_p]),
    "smh_collection_share": (C.c_void_p, [C.c_void_p]),
    "smh_collection_share_to": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "smh_collection_finish": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "smh_collection_compare": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p]),
    "smh_collection_free": (None, [C.c_void_p]),
    "smh_mirror_pack": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_uint32, C.c_void_p, C.c_void_p]),
    "smh_mirror_apply": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_uint32, C.c_void_p, C.c_void_p]),
    "smh_find": (C.c_int, [C.POINTER(C.c_void_p), C.c_uint32, C.c_void_p, C.c_double, C.c_bool, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "smh_most_common": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32), u64p]),
    "smh_index_new": (C.c_void_p, [C.POINTER(C.c_void_p), C.c_uint32]),
    "smh_index_free": (None, [C.c_void_p]),
    "smh_index_drop_dictionary": (None, [C.c_void_p]),
    "smh_index_len": (C.c_uint32, [C.c_void_p]),
    "smh_index_find": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.c_bool, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "smh_index_most_common": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32), u64p]),
    "smh_index_compare": (C.c_int, [C.c_void_p, C.c_void_p, f64p, u64p, u64p, u64p, f64p]),
    "smh_release_workspace": (C.c_int, []),
    "smh_pool_set_limit": (None, [C.c_uint64]),
    "smh_pool_bytes": (C.c_uint64, []),
    "smh_compare_last_stats": (None, [C.POINTER(SmhCompareStats)]),
    "smh_compare_get_tuning": (None, [C.POINTER(SmhCompareTuning)]),
    "smh_compare_set_tuning": (C.c_int, [C.POINTER(SmhCompareTuning)]),
    "smh_synth_dna_dev": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p]),
    "smh_sort_u64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "smh_profile_enable": (None, [C.c_int]),
    "smh_profile_reset": (None, []),
    "smh_profile_get": (C.c_int, [C.c_char_p, f64p, u64p]),
}

_lib = None


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (same SONAME as /opt/rocm's).  Two HIP
    runtimes in one process do not coexist, so when torch is installed its copy is mapped first
    and libsourmash_amd.so's NEEDED libamdhip64.so.7 resolves to it; a later `import torch`
    then finds the same object.  Without torch the system ROCm runtime is used."""
    import importlib.util
    if os.environ.get("SOURMASH_AMD_SYSTEM_HIP") == "1":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise ImportError(
                "libsourmash_amd.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C sourmash-rust_amd/csrc`; there is no pure-Python or CPU fallback." % SO_PATH)
        _share_hip_runtime_with_torch()
        L = C.CDLL(SO_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)  # AttributeError here = the library does not export the ABI
            fn.restype = res
            fn.argtypes = args
        L.sourmash_init()
        _lib = L
    return _lib


def exported_symbols():
    return sorted(_SIGS)
