"""Loader and ctypes prototypes of libphdhip.so (include/phdhip.h). There is no fallback: if the
HIP library is missing or cannot be loaded, importing this module raises."""
import ctypes as C
import os
import subprocess

from .abi import PhdParams

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO_PATH = os.environ.get("PHDHIP_SO") or os.path.join(CSRC, "libphdhip.so")   # (PHDHIP_SO: another build of the same library, for A/B timing)
SOURCES = ["phdhip.hip", "phd_multi.inc", "phd_kernels.h", "phd_correct.h", "phd_sweep.h", "phd_prune.h", "phd_alpha.h", "phd_resample.h", "phd_device.h"]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value", "-Wno-unused-result"]

dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int32)
u8p = C.POINTER(C.c_uint8)


def build(force=False, out=None, defines=()):
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU). `out` / `defines`: a variant build
    (tuning macros of the kernels) next to the product one, selected at run time with PHDHIP_SO."""
    if out:
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        subprocess.check_call([hipcc] + HIPCC_FLAGS + ["-D" + d for d in defines] + ["-o", out, os.path.join(CSRC, "phdhip.hip")], cwd=CSRC)
        return out
    srcs = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(os.path.dirname(HERE), "include", "phdhip.h")]
    if not force and os.path.exists(SO_PATH) and all(os.path.getmtime(SO_PATH) >= os.path.getmtime(s) for s in srcs):
        return SO_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.check_call([hipcc] + HIPCC_FLAGS + ["-o", SO_PATH, os.path.join(CSRC, "phdhip.hip")], cwd=CSRC)
    return SO_PATH


def _share_torch_hip_runtime():
    """PyTorch-ROCm brings its own libamdhip64 (no SONAME, loaded with global scope); libphdhip.so asks for the system's
    libamdhip64.so.7. If the system one initialises first, a later `import torch` finds no GPU ("No HIP GPUs are
    available": two HIP runtimes in one process). Loading PyTorch's runtime first, with global scope, makes libphdhip's
    HIP symbols resolve to it — the arrangement every run has when torch is imported first — whatever the import order.
    Hosts without PyTorch (the C++ and C# ones) are not concerned."""
    if os.environ.get("PHD_SYSTEM_HIP"):
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        rt = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(rt):
            C.CDLL(rt, mode=C.RTLD_GLOBAL)
    except Exception:
        pass   # no PyTorch, or it cannot be located: the system runtime it is


def load():
    _share_torch_hip_runtime()
    if not os.path.exists(SO_PATH):
        raise ImportError("libphdhip.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback for the PHD path)")
    lib = C.CDLL(SO_PATH)
    P = C.c_void_p
    sig = {
        "phd_api_version": (C.c_int, []),
        "phd_default_params": (None, [C.POINTER(PhdParams), C.c_int, C.c_int, C.c_int]),
        "phd_create": (P, [C.POINTER(PhdParams), C.c_int]),
        "phd_create_multi": (P, [C.POINTER(PhdParams), C.POINTER(C.c_int), C.c_int]),
        "phd_create_error": (C.c_char_p, []),
        "phd_destroy": (None, [P]),
        "phd_last_error": (C.c_char_p, [P]),
        "phd_reset": (C.c_int, [P, C.c_int, dp, dp, dp, dp, C.c_int]),
        "phd_set_poses": (C.c_int, [P, dp, C.c_int]),
        "phd_set_weights": (C.c_int, [P, dp, C.c_int]),
        "phd_set_map": (C.c_int, [P, C.c_int, dp, dp, dp, C.c_int]),
        "phd_upload_state_soa": (C.c_int, [P, C.c_int, C.c_int, dp, ip, dp, dp]),
        "phd_download_state_soa": (C.c_int, [P, C.c_int, dp, ip, dp, dp]),
        "phd_slam_update": (C.c_int, [P, dp, C.c_int, C.c_uint8, C.c_double]),
        "phd_set_measurements": (C.c_int, [P, dp, C.c_int]),
        "phd_step_async": (C.c_int, [P, C.c_uint8, C.c_double]),
        "phd_sync": (C.c_int, [P]),
        "phd_set_frozen": (C.c_int, [P, C.c_uint8]),
        "phd_set_all_pairs": (C.c_int, [P, C.c_uint8]),
        "phd_set_association_workspace": (C.c_int, [P, C.c_int64]),
        "phd_update_motion": (C.c_int, [P, dp, dp, C.c_int, C.c_uint8]),
        "phd_quasi_set_loglik": (C.c_int, [P, dp, C.c_int, dp, C.c_int, dp, C.c_int, dp]),
        "phd_quasi_set_loglik_grad": (C.c_int, [P, dp, C.c_int, dp, C.c_int, dp, C.c_int, C.c_int, dp, dp]),
        "phd_test_pairing": (C.c_int, [P, dp, C.c_int, C.c_int, C.c_int, C.c_int, ip, dp, C.POINTER(C.c_int)]),
        "phd_set_split": (C.c_int, [P, C.c_int]),
        "phd_weights": (dp, [P, ip]),
        "phd_best_particle": (C.c_int, [P]),
        "phd_poses": (dp, [P, ip]),
        "phd_particle_count": (C.c_int, [P]),
        "phd_map": (C.c_int, [P, C.c_int, ip, C.POINTER(dp), C.POINTER(dp), C.POINTER(dp)]),
        "phd_resample_sources": (ip, [P, ip, u8p]),
        "phd_stage_run": (C.c_int, [P, dp, C.c_int, C.c_uint8]),
        "phd_stage_map": (C.c_int, [P, C.c_int, C.c_int, ip, C.POINTER(dp), C.POINTER(dp), C.POINTER(dp)]),
        "phd_stage_alpha": (dp, [P, ip]),
        "phd_stage_setloglik": (dp, [P, ip]),
        "phd_resample": (C.c_int, [P, dp, C.c_int, C.c_double, ip, ip]),
        "phd_particle_depleted": (C.c_int, [P, dp, C.c_int, u8p]),
        "phd_step_local_async": (C.c_int, [P, C.c_uint8]),
        "phd_device_local_weights": (C.c_void_p, [P]),
        "phd_device_global_weights": (C.c_void_p, [P, C.c_int]),
        "phd_step_global_async": (C.c_int, [P, C.c_int, C.c_int, C.c_double]),
        "phd_migration_plan": (C.c_int, [P, C.c_int, C.c_int, ip, ip]),
        "phd_plan_migration": (C.c_int, [ip, C.c_int, C.c_int, C.c_int, ip, ip, ip, ip]),
        "phd_test_migration_plan": (C.c_int, [P, ip, C.c_int, C.c_int, C.c_int, C.c_int, ip, ip, ip, ip, ip, ip, ip, ip, ip]),
        "phd_multi_report": (C.c_int, [P, dp, u8p, ip]),
        "phd_last_resampled": (C.c_int, [P]),
        "phd_migration_send_buffer": (C.c_void_p, [P, C.POINTER(C.c_int64)]),
        "phd_migration_recv_buffer": (C.c_void_p, [P]),
        "phd_migration_pack_async": (C.c_int, [P]),
        "phd_migration_unpack_async": (C.c_int, [P]),
        "phd_device_gather_buffer": (C.c_void_p, [P, C.c_int]),
        "phd_step_global_device_async": (C.c_int, [P, C.c_int, C.c_int, C.c_double, C.c_uint8]),
        "phd_migration_ipc_export": (C.c_int, [P, C.c_void_p, C.POINTER(C.c_int64)]),
        "phd_migration_ipc_open": (C.c_int, [P, C.c_void_p, C.c_int, C.c_int]),
        "phd_migration_set_peers": (C.c_int, [P, C.POINTER(C.c_void_p), C.c_int, C.c_int]),
        "phd_migration_recv_is_finegrained": (C.c_int, [P]),
        "phd_migration_push_async": (C.c_int, [P]),
        "phd_migration_set_landing": (C.c_int, [P, C.c_int]),
        "phd_stream": (C.c_void_p, [P]),
        "phd_set_stream": (C.c_int, [P, C.c_void_p, C.c_uint8]),
        "phd_timing_reset": (C.c_int, [P, C.c_uint8]),
        "phd_last_timings": (C.c_int, [P, C.POINTER(C.POINTER(C.c_char_p)), C.POINTER(dp)]),
        "phd_last_timing_counts": (C.c_int, [P, C.POINTER(C.POINTER(C.c_int))]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)   # raises AttributeError if the library does not export a declared symbol
        fn.restype, fn.argtypes = res, args
    return lib


EXPORTS = ["phd_api_version", "phd_default_params", "phd_create", "phd_create_multi", "phd_create_error", "phd_destroy", "phd_last_error",
           "phd_reset", "phd_set_poses", "phd_set_weights", "phd_set_map", "phd_slam_update", "phd_set_measurements",
           "phd_step_async", "phd_sync", "phd_set_frozen", "phd_set_all_pairs", "phd_set_association_workspace", "phd_update_motion", "phd_quasi_set_loglik", "phd_quasi_set_loglik_grad", "phd_test_pairing", "phd_set_split", "phd_weights", "phd_best_particle", "phd_poses",
           "phd_particle_count", "phd_map", "phd_resample_sources", "phd_stage_run", "phd_stage_map", "phd_stage_alpha",
           "phd_stage_setloglik", "phd_resample", "phd_particle_depleted", "phd_step_local_async",
           "phd_device_local_weights", "phd_device_global_weights", "phd_step_global_async", "phd_migration_plan",
           "phd_plan_migration", "phd_test_migration_plan", "phd_multi_report", "phd_last_resampled", "phd_migration_send_buffer", "phd_migration_recv_buffer", "phd_migration_pack_async",
           "phd_migration_unpack_async", "phd_device_gather_buffer", "phd_step_global_device_async", "phd_migration_ipc_export", "phd_migration_ipc_open",
           "phd_migration_set_peers", "phd_migration_recv_is_finegrained", "phd_migration_push_async", "phd_migration_set_landing", "phd_stream", "phd_set_stream", "phd_timing_reset", "phd_last_timings", "phd_last_timing_counts", "phd_upload_state_soa",
           "phd_download_state_soa"]
