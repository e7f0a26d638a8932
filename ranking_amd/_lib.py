"""Builds and binds ``libtfr_hip.so`` (the C ABI declared in include/tfr_hip.h).

There is deliberately NO fallback: if the shared library cannot be built or
loaded, or a tensor is not resident on a HIP device, the product path raises.
"""
from __future__ import annotations

import ctypes
import os
import shutil
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
INCLUDE = os.path.join(os.path.dirname(_HERE), 'include')
LIB_PATH = os.path.join(CSRC, 'libtfr_hip.so')
STAMP_PATH = LIB_PATH + '.stamp'                           # fingerprint of what LIB_PATH was built from
PROF_LIB_PATH = os.path.join(CSRC, 'libtfr_hip_prof.so')     # developer aid: -DTFR_PROFILE_STAMPS build
SOURCES = ['sort_metrics.hip', 'approx_ndcg.hip', 'pairwise.hip', 'softmax_gumbel.hip', 'tower.hip', 'listwise.hip',
           'neural_sort.hip', 'pointwise.hip', 'groupwise.hip', 'gemm_f32.hip']
HIPCC_FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared',
               '-fvisibility=default']

_lock = threading.Lock()
_lib = None

c_f32p = ctypes.c_void_p
_WS = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p]        # ... void* workspace, long workspace_bytes, void* stream
_SIGNATURES = {
    # name: (restype, argtypes)
    'tfr_hip_abi_version': (ctypes.c_int, []),
    'tfr_sort_ranks_f32': (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_int] * 2 + [ctypes.c_void_p] * 3),
    'tfr_ndcg_metric_f32': (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_int] + [ctypes.c_void_p] * 3
                            + [ctypes.POINTER(ctypes.c_int32)] + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 2
                            + [ctypes.c_uint32, ctypes.c_void_p]),
    'tfr_rank_metric_f32': (ctypes.c_int, [ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int] + [ctypes.c_void_p] * 3
                            + [ctypes.POINTER(ctypes.c_int32)] + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 2
                            + [ctypes.c_uint32] + _WS),
    'tfr_mrr_metric_f32': (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_int] + [ctypes.c_void_p]
                           + [ctypes.POINTER(ctypes.c_int32)] + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 2
                           + [ctypes.c_uint32, ctypes.c_void_p]),
    'tfr_approx_ndcg_f32': (ctypes.c_int, [ctypes.c_void_p] * 5 + [ctypes.c_int] * 2 + [ctypes.c_float]
                            + [ctypes.c_int] + [ctypes.c_void_p] * 5),
    'tfr_list_order_i32': (ctypes.c_int, [ctypes.c_void_p] * 2 + [ctypes.c_int] * 2 + [ctypes.c_void_p] * 3),
    'tfr_list_order_interleaved_i32': (ctypes.c_int, [ctypes.c_void_p] * 2 + [ctypes.c_int] * 2 + [ctypes.c_void_p] * 2),
    'tfr_grid_sum_state_ints': (ctypes.c_int, []),
    'tfr_approx_ndcg_sum_f32': (ctypes.c_int, [ctypes.c_void_p] * 5 + [ctypes.c_int] * 2 + [ctypes.c_float]
                                + [ctypes.c_int] + [ctypes.c_void_p] * 7),
    'tfr_approx_mrr_f32': (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_int] * 2 + [ctypes.c_float]
                           + [ctypes.c_void_p] * 5),
    'tfr_list_workspace_bytes': (ctypes.c_long, [ctypes.c_int] * 2),
    'tfr_list_mle_f32': (ctypes.c_int, [ctypes.c_void_p] * 5 + [ctypes.c_int] * 2 + [ctypes.c_float]
                         + [ctypes.c_void_p] * 2 + [ctypes.c_uint32] + _WS),
    'tfr_unique_softmax_f32': (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_int] * 2 + [ctypes.c_float]
                               + [ctypes.c_void_p] * 2 + _WS),
    'tfr_metric_list_weights_f32': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    'tfr_div_metric_f32': (ctypes.c_int, [ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int]
                           + [ctypes.c_void_p] * 2 + [ctypes.c_float] + [ctypes.c_void_p] + [ctypes.c_int] * 4
                           + [ctypes.c_void_p] * 2 + [ctypes.c_uint32] + _WS),
    'tfr_pointwise_loss_f32': (ctypes.c_int, [ctypes.c_int] + [ctypes.c_void_p] * 5 + [ctypes.c_int] * 2
                               + [ctypes.c_float] + [ctypes.c_void_p] * 5),
    'tfr_circle_loss_f32': (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_int] * 2 + [ctypes.c_float] * 2
                            + [ctypes.c_int] + [ctypes.c_void_p] * 3 + _WS),
    'tfr_neural_sort_loss_f32': (ctypes.c_int, [ctypes.c_int] + [ctypes.c_void_p] * 5 + [ctypes.c_int] * 2
                                 + [ctypes.c_float] + [ctypes.c_void_p] * 2 + _WS),
    'tfr_pairwise_logistic_f32': (ctypes.c_int, [ctypes.c_void_p] * 5 + [ctypes.c_int] * 2 + [ctypes.c_float]
                                  + [ctypes.c_int] * 2 + [ctypes.c_void_p] * 2 + [ctypes.c_int] * 2
                                  + [ctypes.c_float] + [ctypes.c_void_p] * 5),
    'tfr_pairwise_loss_f32': (ctypes.c_int, [ctypes.c_int] + [ctypes.c_void_p] * 5 + [ctypes.c_int] * 2
                              + [ctypes.c_float] + [ctypes.c_int] * 2 + [ctypes.c_void_p] * 2 + [ctypes.c_int] * 2
                              + [ctypes.c_float] + [ctypes.c_void_p] * 6 + [ctypes.c_uint32, ctypes.c_void_p]),
    'tfr_softmax_loss_f32': (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_int] * 5 + [ctypes.c_void_p] * 2
                             + [ctypes.c_int] * 2 + [ctypes.c_float] + [ctypes.c_void_p] * 4),
    'tfr_poly1_softmax_loss_f32': (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_int] * 5 + [ctypes.c_void_p] * 2
                                   + [ctypes.c_int] * 2 + [ctypes.c_float] * 2 + [ctypes.c_void_p] * 4),
    'tfr_gumbel_sample_f32': (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_uint64] * 2 + [ctypes.c_int] * 3
                              + [ctypes.c_float] + [ctypes.c_void_p] * 2),
    'tfr_gumbel_sample_bwd_f32': (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_int] * 3 + [ctypes.c_float]
                                  + [ctypes.c_void_p] * 2),
    'tfr_gumbel_sample_step_f32': (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_uint64] * 2 + [ctypes.c_void_p]
                                   + [ctypes.c_int] * 3 + [ctypes.c_float] + [ctypes.c_void_p] * 3),
    'tfr_gumbel_sample_bwd_step_f32': (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_int] * 3 + [ctypes.c_float]
                                       + [ctypes.c_void_p] * 3),
    # scorer tower (tower.hip)
    'tfr_tower_bn_bwd_coeffs': (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_long]
                                + [ctypes.c_void_p] * 2),
    'tfr_tower_input_stats_f32': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                                 ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    'tfr_tower_cast_gather_f32_bf16': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_long] + [ctypes.c_int] * 3
                                       + [ctypes.c_void_p] * 5),
    'tfr_tower_input_stats_bf16': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                                  ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    'tfr_tower_cast_gather_bf16_bf16': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_long] + [ctypes.c_int] * 3
                                        + [ctypes.c_void_p] * 5),
    'tfr_tower_cast_f32_bf16': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_long] + [ctypes.c_int] * 3
                                + [ctypes.c_void_p] * 4),
    'tfr_tower_weight_cast': (ctypes.c_int, [ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p] * 2),
    'tfr_tower_gemm_bf16': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_long] * 3 + [ctypes.c_int] * 4
                            + [ctypes.c_void_p] * 3 + [ctypes.c_int] + [ctypes.c_void_p] * 2 + [ctypes.c_long]
                            + [ctypes.c_void_p] * 7),
    'tfr_tower_gemm_bf16_aout': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_long] * 3 + [ctypes.c_int] * 4
                                 + [ctypes.c_void_p] * 3 + [ctypes.c_int] + [ctypes.c_void_p] * 2 + [ctypes.c_long]
                                 + [ctypes.c_void_p] * 6 + [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p]),
    'tfr_tower_gemm_writes_operand': (ctypes.c_int, [ctypes.c_int] * 3),
    'tfr_tower_gemm_stats_rows': (ctypes.c_int, [ctypes.c_int]),
    'tfr_tower_reduce_scratch_rows': (ctypes.c_int, [ctypes.c_int]),
    'tfr_tower_bn_finalize': (ctypes.c_int, [ctypes.c_void_p] + [ctypes.c_int] * 2 + [ctypes.c_long]
                              + [ctypes.c_void_p] * 2 + [ctypes.c_float] * 2 + [ctypes.c_void_p] * 8),
    'tfr_tower_reduce_partials': (ctypes.c_int, [ctypes.c_void_p] + [ctypes.c_int] * 2 + [ctypes.c_void_p] * 3),
    'tfr_tower_reduce_partials_coeffs': (ctypes.c_int, [ctypes.c_void_p] + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 5
                                         + [ctypes.c_long] + [ctypes.c_void_p] * 2),
    'tfr_tower_reduce_partials_serves_db': (ctypes.c_int, [ctypes.c_int] * 2),
    'tfr_tower_reduce_partials_coeffs_db': (ctypes.c_int, [ctypes.c_void_p] + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 5
                                            + [ctypes.c_long] + [ctypes.c_void_p] * 2 + [ctypes.c_long, ctypes.c_int]
                                            + [ctypes.c_void_p] * 2),
    'tfr_tower_out_f32': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_long] + [ctypes.c_int] * 3
                          + [ctypes.c_void_p] * 4 + [ctypes.c_int] + [ctypes.c_void_p] * 3),
    'tfr_tower_out_bwd': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_long] + [ctypes.c_int] * 3
                          + [ctypes.c_void_p] * 6 + [ctypes.c_int] + [ctypes.c_void_p, ctypes.c_long]
                          + [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    'tfr_tower_weight_cast_batch': (ctypes.c_int, [ctypes.c_void_p] * 6 + [ctypes.c_int, ctypes.c_void_p]),
    'tfr_tower_weight_cast_batch_step': (ctypes.c_int, [ctypes.c_void_p] * 6 + [ctypes.c_int] + [ctypes.c_void_p] * 3),
    'tfr_tower_multi_add': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    'tfr_flatten_row_index': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    'tfr_tower_out_bwd2': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_long] + [ctypes.c_int] * 3
                           + [ctypes.c_void_p] * 6 + [ctypes.c_int] + [ctypes.c_void_p, ctypes.c_long]
                           + [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    'tfr_tower_bn_bwd_apply': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_long] * 2 + [ctypes.c_int] * 2
                               + [ctypes.c_void_p] * 2),
    'tfr_tower_wgrad_bf16': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_long] * 2 + [ctypes.c_int] * 4
                             + [ctypes.c_void_p] * 3 + [ctypes.c_long, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    'tfr_tower_slab_reduce': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_void_p,
                                             ctypes.c_int, ctypes.c_void_p]),
    'tfr_tower_slab_reduce_cols': (ctypes.c_int, [ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    # fp32 Dense on the matrix cores (gemm_f32.hip)
    'tfr_tower_gemm_f32': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_long, ctypes.c_int] * 2 + [ctypes.c_void_p, ctypes.c_long]
                           + [ctypes.c_int] * 3 + [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    'tfr_tower_gemm_f32_splits': (ctypes.c_int, [ctypes.c_int] * 3),
    'tfr_tower_colsum_f32': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int]
                             + [ctypes.c_void_p] * 3),
    'tfr_tower_colsum_rows': (ctypes.c_int, [ctypes.c_int]),
    'tfr_list_dot_f32': (ctypes.c_int, [ctypes.c_void_p] * 2 + [ctypes.c_int] + [ctypes.c_void_p] * 2),
    # the reduced scalar from the loss launch itself (round 5): the plain entry point's arguments + sum / scratch / ticket
    'tfr_softmax_loss_sum_f32': (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_int] * 5 + [ctypes.c_void_p] * 2
                                 + [ctypes.c_int] * 2 + [ctypes.c_float] * 2 + [ctypes.c_void_p] * 7),
    'tfr_softmax_sum_contributors': (ctypes.c_int, [ctypes.c_int] * 6),
    'tfr_pairwise_loss_sum_f32': (ctypes.c_int, [ctypes.c_int] + [ctypes.c_void_p] * 5 + [ctypes.c_int] * 2
                                  + [ctypes.c_float] + [ctypes.c_int] * 2 + [ctypes.c_void_p] * 2 + [ctypes.c_int] * 2
                                  + [ctypes.c_float] + [ctypes.c_void_p] * 8 + [ctypes.c_uint32, ctypes.c_void_p]),
    'tfr_list_mle_sum_f32': (ctypes.c_int, [ctypes.c_void_p] * 5 + [ctypes.c_int] * 2 + [ctypes.c_float]
                             + [ctypes.c_void_p] * 4 + [ctypes.c_uint32] + _WS),
    'tfr_unique_softmax_sum_f32': (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_int] * 2 + [ctypes.c_float]
                                   + [ctypes.c_void_p] * 4 + _WS),
    'tfr_pointwise_loss_sum_f32': (ctypes.c_int, [ctypes.c_int] + [ctypes.c_void_p] * 5 + [ctypes.c_int] * 2
                                   + [ctypes.c_float] + [ctypes.c_void_p] * 7),
    # groupwise scoring (groupwise.hip)
    'tfr_group_indices_i32': (ctypes.c_int, [ctypes.c_void_p] * 2 + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 3),
    'tfr_group_gather_cast_f32_bf16': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p]
                                       + [ctypes.c_int] * 6 + [ctypes.c_void_p] * 2),
    'tfr_group_scatter_avg_f32': (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_int] * 4 + [ctypes.c_void_p] * 3),
    'tfr_group_scatter_avg_bwd_f32': (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_int] * 4
                                      + [ctypes.c_void_p] * 2),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)
ABI_VERSION = 2                        # = TFR_HIP_ABI_VERSION of include/tfr_hip.h (tests/test_host_logic.py compares them)


class TfrHipError(RuntimeError):
    pass


def sources_present():
    return [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _deps():
    deps = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(('.hip', '.h'))]
    deps.append(os.path.join(INCLUDE, 'tfr_hip.h'))
    return [d for d in deps if os.path.exists(d)]


def _fingerprint() -> str:
    """sha256 over the sources, the public header and the compiler flags: what the in-tree .so was built from."""
    import hashlib
    h = hashlib.sha256(' '.join(HIPCC_FLAGS).encode())
    for d in _deps():
        h.update(os.path.basename(d).encode())
        with open(d, 'rb') as f:
            h.update(f.read())
    return h.hexdigest()


def _stale():
    """Content, not timestamps, decides (a snapshot copied to a GPU box keeps no useful mtimes): the stamp file next
    to the library holds the fingerprint it was built from.  Without a stamp (an older build) mtimes decide."""
    if not os.path.exists(LIB_PATH):
        return True
    try:
        with open(STAMP_PATH) as f:
            return f.read().strip() != _fingerprint()
    except OSError:
        t = os.path.getmtime(LIB_PATH)
        return any(os.path.getmtime(d) > t for d in _deps())


OBJ_DIR = os.path.join(CSRC, '_obj')                     # per-source objects + their fingerprints (git-ignored)
COMPILE_FLAGS = [f for f in HIPCC_FLAGS if f != '-shared']


class _BuildLock:
    """Inter-process lock around stale-check + build: under torchrun every rank may find the library stale at the
    same moment; one compiles, the others wait and then see a fresh stamp."""

    def __init__(self, path):
        self._path, self._fd = path, None

    def __enter__(self):
        import fcntl
        self._fd = os.open(self._path, os.O_CREAT | os.O_RDWR, 0o644)
        fcntl.flock(self._fd, fcntl.LOCK_EX)
        return self

    def __exit__(self, *exc):
        import fcntl
        fcntl.flock(self._fd, fcntl.LOCK_UN)
        os.close(self._fd)


def _source_fingerprint(src: str) -> str:
    """What one object file depends on: its source, the shared headers and the compile flags."""
    import hashlib
    h = hashlib.sha256(' '.join(COMPILE_FLAGS).encode())
    deps = [os.path.join(CSRC, src)] + [d for d in _deps() if d.endswith('.h')]
    for d in deps:
        h.update(os.path.basename(d).encode())
        with open(d, 'rb') as f:
            h.update(f.read())
    return h.hexdigest()


def _compile_one(hipcc: str, src: str, verbose: bool) -> str:
    obj = os.path.join(OBJ_DIR, src + '.o')
    stamp = obj + '.stamp'
    fp = _source_fingerprint(src)
    try:
        with open(stamp) as f:
            if f.read().strip() == fp and os.path.exists(obj):
                return obj
    except OSError:
        pass
    tmp = '%s.%d.tmp' % (obj, os.getpid())
    cmd = [hipcc] + COMPILE_FLAGS + ['-I', INCLUDE, '-c', os.path.join(CSRC, src), '-o', tmp]
    if verbose:
        print(' '.join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise TfrHipError('hipcc failed on %s:\n%s\n%s' % (src, res.stdout, res.stderr))
    os.replace(tmp, obj)
    with open(stamp, 'w') as f:
        f.write(fp + '\n')
    return obj


def build(force: bool = False, verbose: bool = False) -> str:
    """Compiles every HIP translation unit for gfx950 (one object per source, in parallel, re-using the objects
    whose source / headers / flags did not change) and links them into the in-tree libtfr_hip.so."""
    with _lock:
        os.makedirs(OBJ_DIR, exist_ok=True)
        with _BuildLock(os.path.join(OBJ_DIR, '.lock')):
            if not force and not _stale():
                return LIB_PATH
            hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
            if not os.path.exists(hipcc):
                raise TfrHipError('hipcc not found: cannot build %s' % LIB_PATH)
            if force:
                for f in os.listdir(OBJ_DIR):
                    if f.endswith('.stamp'):
                        os.remove(os.path.join(OBJ_DIR, f))
            from concurrent.futures import ThreadPoolExecutor
            srcs = sources_present()
            with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 1)) as pool:
                objs = list(pool.map(lambda s_: _compile_one(hipcc, s_, verbose), srcs))
            tmp = '%s.%d.tmp' % (LIB_PATH, os.getpid())
            cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-fvisibility=default'] + objs + ['-o', tmp]
            if verbose:
                print(' '.join(cmd))
            res = subprocess.run(cmd, capture_output=True, text=True)
            if res.returncode != 0:
                raise TfrHipError('hipcc link failed:\n%s\n%s' % (res.stdout, res.stderr))
            os.replace(tmp, LIB_PATH)
            with open(STAMP_PATH, 'w') as f:
                f.write(_fingerprint() + '\n')
            return LIB_PATH


def build_profiling(verbose: bool = False) -> str:
    """Developer aid: the same sources with -DTFR_PROFILE_STAMPS (in-kernel s_memtime stamps);
    used by tools/phase_profile.py only, never by the product path."""
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    srcs = [os.path.join(CSRC, s) for s in sources_present()]
    cmd = [hipcc] + HIPCC_FLAGS + ['-DTFR_PROFILE_STAMPS', '-I', INCLUDE] + srcs + ['-o', PROF_LIB_PATH]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise TfrHipError('hipcc failed:\n%s\n%s' % (res.stdout, res.stderr))
    return PROF_LIB_PATH


def load():
    """Returns the ctypes handle; builds first when the in-tree .so is missing/stale
    and hipcc is available.  Raises TfrHipError otherwise -- never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if _stale():
        hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
        if os.path.exists(hipcc):
            build()
        elif not os.path.exists(LIB_PATH):
            raise TfrHipError('%s is missing and hipcc is unavailable; the HIP extension is '
                              'mandatory (no CPU fallback).' % LIB_PATH)
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:   # pragma: no cover
        raise TfrHipError('cannot load %s: %s' % (LIB_PATH, e))
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise TfrHipError('%s does not export %s' % (LIB_PATH, name))
        fn.restype = res
        fn.argtypes = args
    got = lib.tfr_hip_abi_version()
    if got != ABI_VERSION:            # a stale / foreign libtfr_hip.so: struct layouts and argument lists may differ
        raise TfrHipError('%s reports ABI version %d, this binding was written against %d (include/tfr_hip.h '
                          'TFR_HIP_ABI_VERSION): rebuild it (__graft_entry__.build())' % (LIB_PATH, got, ABI_VERSION))
    _lib = lib
    return lib


_TRACE_CALLS = bool(os.environ.get('TFR_SYNC_EVERY_CALL'))     # developer switch: localise an asynchronous GPU fault


def check(code: int, what: str):
    if _TRACE_CALLS:                                          # name first, then wait: the last name printed is the culprit
        import sys
        import torch
        sys.stderr.write('[tfr] %s\n' % what)
        sys.stderr.flush()
        if not torch.cuda.is_current_stream_capturing():
            torch.cuda.synchronize()
    if code == 0:
        return
    if code == -1:
        raise ValueError('%s: invalid argument (TFR_EINVAL)' % what)
    if code == -2:
        raise ValueError('%s: list_size exceeds what one workgroup can hold in LDS '
                         '(TFR_ETOOLARGE)' % what)
    raise TfrHipError('%s: hipError_t %d' % (what, code))
