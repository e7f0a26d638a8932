"""Loader of the plain-C restatement of the headline path (oracle/approx_ndcg_c.c).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__ and bench.py's cpu_baseline leg, never by ranking_amd/.

build() compiles with gcc into oracle/_build/ (git-ignored, travels with gpurun):
  libapprox_ndcg_f64.so        strict fp64 arbiter            tfr_c_approx_ndcg_f64
  libapprox_ndcg_f32.so        -Ofast -march=native float     tfr_c_approx_ndcg_f32_fast
  libpairwise_softmax_f64.so   pairwise_softmax_c.c (fp64)    tfr_c_pairwise_logistic_ndcg_f64, tfr_c_softmax_f64
  liblistwise_f64.so           listwise_c.c (fp64)            tfr_c_list_mle_f64, tfr_c_unique_softmax_f64, tfr_c_circle_f64
"""
import ctypes
import hashlib
import os
import shutil
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'approx_ndcg_c.c')
SRC_PS = os.path.join(HERE, 'pairwise_softmax_c.c')
SRC_LW = os.path.join(HERE, 'listwise_c.c')
OUT = os.path.join(HERE, '_build')
_VARIANTS = {   # key: (library, entry point, flags, source)
    'f64': ('libapprox_ndcg_f64.so', 'tfr_c_approx_ndcg_f64', ['-O2', '-fno-fast-math'], SRC),
    'f32_fast': ('libapprox_ndcg_f32.so', 'tfr_c_approx_ndcg_f32_fast', ['-DTFR_C_FLOAT', '-Ofast', '-march=native'], SRC),
    'ps_f64': ('libpairwise_softmax_f64.so', 'tfr_c_softmax_f64', ['-O2', '-fno-fast-math'], SRC_PS),
    'lw_f64': ('liblistwise_f64.so', 'tfr_c_list_mle_f64', ['-O2', '-fno-fast-math'], SRC_LW),
}
_handles = {}


def _host_isa():
    """-march=native output is only valid on the CPU it was built on: the ISA flag set is part of the fingerprint, so
    a snapshot that travels to another host (the GPU box) rebuilds there instead of dying on an illegal instruction."""
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith(('flags', 'Features')):
                    return hashlib.sha256(line.encode()).hexdigest()[:16]
    except OSError:
        pass
    return 'unknown'


def _fingerprint(flags, src=SRC):
    with open(src, 'rb') as f:
        body = f.read()
    isa = _host_isa() if '-march=native' in flags else ''
    return hashlib.sha256(body + ' '.join(flags).encode() + isa.encode()).hexdigest()


def build(force=False):
    """gcc -shared -fPIC -fopenmp, once per variant; rebuilt when the source or the flags change."""
    os.makedirs(OUT, exist_ok=True)
    gcc = shutil.which('gcc')
    paths = {}
    for key, (lib, _sym, flags, src) in _VARIANTS.items():
        path, stamp = os.path.join(OUT, lib), os.path.join(OUT, lib + '.stamp')
        fp = _fingerprint(flags, src)
        fresh = os.path.exists(path) and os.path.exists(stamp) and open(stamp).read().strip() == fp
        if force or not fresh:
            if gcc is None:
                raise RuntimeError('gcc not found: cannot build %s' % path)
            cmd = [gcc, '-std=c11', '-shared', '-fPIC', '-fopenmp'] + flags + [src, '-o', path + '.tmp', '-lm']
            res = subprocess.run(cmd, capture_output=True, text=True)
            if res.returncode != 0:
                raise RuntimeError('gcc failed:\n%s\n%s' % (res.stdout, res.stderr))
            os.replace(path + '.tmp', path)
            with open(stamp, 'w') as f:
                f.write(fp + '\n')
        paths[key] = path
    return paths


def _fn(variant):
    if variant not in _handles:
        path = build()[variant]
        lib = ctypes.CDLL(path)
        fn = getattr(lib, _VARIANTS[variant][1])
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_int, ctypes.c_float] + [ctypes.c_void_p] * 3
        _handles[variant] = fn
    return _handles[variant]


def approx_ndcg(logits, labels, mask=None, temperature=0.1, want_grad=True, variant='f64'):
    """(loss [B], weight [B], dlogits [B, L] | None) as numpy fp32; per-list loss = -ApproxNDCG, weight = 1{sum
    label > 0}, dlogits = d(sum_b loss_b) / d logits (losses_impl.py:1579-1603 with :77-167)."""
    logits = np.ascontiguousarray(np.asarray(logits, dtype=np.float32))
    labels = np.ascontiguousarray(np.asarray(labels, dtype=np.float32))
    B, L = logits.shape
    m = None if mask is None else np.ascontiguousarray(np.asarray(mask).astype(np.uint8))
    loss = np.empty(B, dtype=np.float32)
    weight = np.empty(B, dtype=np.float32)
    grad = np.empty((B, L), dtype=np.float32) if want_grad else None
    rc = _fn(variant)(logits.ctypes.data, labels.ctypes.data, None if m is None else m.ctypes.data, B, L,
                      float(temperature), loss.ctypes.data, weight.ctypes.data,
                      None if grad is None else grad.ctypes.data)
    if rc != 0:
        raise ValueError('approx_ndcg_c: invalid argument')
    return loss, weight, grad


def threads():
    """omp_get_max_threads() of the C loops (what `cores` means for a timing of them)."""
    lib = ctypes.CDLL(build()['f64'])
    lib.tfr_c_threads.restype = ctypes.c_int
    return int(lib.tfr_c_threads())


def _prep(logits, labels, mask):
    logits = np.ascontiguousarray(np.asarray(logits, dtype=np.float32))
    labels = np.ascontiguousarray(np.asarray(labels, dtype=np.float32))
    m = None if mask is None else np.ascontiguousarray(np.asarray(mask).astype(np.uint8))
    return logits, labels, m


def pairwise_logistic_ndcg(logits, labels, mask=None, temperature=1.0, want_grad=True):
    """(out [B], dlogits [B, L] | None): out[b] = sum_ij w_ij * logistic(s_i - s_j) with the Keras NDCGLambdaWeight()
    pair weights (losses_impl.py:255-369, 483-537, 863-940); fp64 inside.  mask None, or equal to labels >= 0."""
    logits, labels, m = _prep(logits, labels, mask)
    B, L = logits.shape
    lib = ctypes.CDLL(build()['ps_f64'])
    fn = lib.tfr_c_pairwise_logistic_ndcg_f64
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_int, ctypes.c_float] + [ctypes.c_void_p] * 2
    out = np.empty(B, dtype=np.float32)
    grad = np.empty((B, L), dtype=np.float32) if want_grad else None
    if fn(logits.ctypes.data, labels.ctypes.data, None if m is None else m.ctypes.data, B, L, float(temperature),
          out.ctypes.data, None if grad is None else grad.ctypes.data) != 0:
        raise ValueError('pairwise_logistic_ndcg_c: invalid argument')
    return out, grad


def softmax(logits, labels, mask=None, temperature=1.0, want_grad=True):
    """(loss [B], weight [B], dlogits | None) of SoftmaxLoss without lambda weight (losses_impl.py:1119-1197)."""
    logits, labels, m = _prep(logits, labels, mask)
    B, L = logits.shape
    lib = ctypes.CDLL(build()['ps_f64'])
    fn = lib.tfr_c_softmax_f64
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_int, ctypes.c_float] + [ctypes.c_void_p] * 3
    loss = np.empty(B, dtype=np.float32)
    weight = np.empty(B, dtype=np.float32)
    grad = np.empty((B, L), dtype=np.float32) if want_grad else None
    if fn(logits.ctypes.data, labels.ctypes.data, None if m is None else m.ctypes.data, B, L, float(temperature),
          loss.ctypes.data, weight.ctypes.data, None if grad is None else grad.ctypes.data) != 0:
        raise ValueError('softmax_c: invalid argument')
    return loss, weight, grad


def ndcg_mrr(predictions, labels, mask=None, topn=None):
    """(ndcg [B], mrr [B]) per list, unweighted (metrics_impl.py:429-459, 631-670); fp64 inside."""
    predictions, labels, m = _prep(predictions, labels, mask)
    B, L = predictions.shape
    lib = ctypes.CDLL(build()['ps_f64'])
    fn = lib.tfr_c_ndcg_mrr_f64
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 2
    ndcg = np.empty(B, dtype=np.float32)
    mrr = np.empty(B, dtype=np.float32)
    if fn(predictions.ctypes.data, labels.ctypes.data, None if m is None else m.ctypes.data, B, L,
          0 if topn is None else int(topn), ndcg.ctypes.data, mrr.ctypes.data) != 0:
        raise ValueError('ndcg_mrr_c: invalid argument')
    return ndcg, mrr


def list_mle(logits, labels, mask=None, pos_weight=None, temperature=1.0, want_grad=True):
    """(loss [B], dlogits [B, L] | None) of ListMLELoss (losses_impl.py:1541-1576; pos_weight[p] = the
    ListMLELambdaWeight rank discount of position p + 1, :457-480); fp64 inside, O(L^2) loops."""
    logits, labels, m = _prep(logits, labels, mask)
    B, L = logits.shape
    pw = None if pos_weight is None else np.ascontiguousarray(np.asarray(pos_weight, dtype=np.float32))
    lib = ctypes.CDLL(build()['lw_f64'])
    fn = lib.tfr_c_list_mle_f64
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_int, ctypes.c_float] + [ctypes.c_void_p] * 2
    loss = np.empty(B, dtype=np.float32)
    grad = np.empty((B, L), dtype=np.float32) if want_grad else None
    if fn(logits.ctypes.data, labels.ctypes.data, None if m is None else m.ctypes.data,
          None if pw is None else pw.ctypes.data, B, L, float(temperature), loss.ctypes.data,
          None if grad is None else grad.ctypes.data) != 0:
        raise ValueError('list_mle_c: invalid argument')
    return loss, grad


def unique_softmax(logits, labels, mask=None, temperature=1.0, want_grad=True):
    """(loss [B], dlogits [B, L] | None) of UniqueSoftmaxLoss (losses_impl.py:1250-1281); fp64 inside."""
    logits, labels, m = _prep(logits, labels, mask)
    B, L = logits.shape
    lib = ctypes.CDLL(build()['lw_f64'])
    fn = lib.tfr_c_unique_softmax_f64
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_int, ctypes.c_float] + [ctypes.c_void_p] * 2
    loss = np.empty(B, dtype=np.float32)
    grad = np.empty((B, L), dtype=np.float32) if want_grad else None
    if fn(logits.ctypes.data, labels.ctypes.data, None if m is None else m.ctypes.data, B, L, float(temperature),
          loss.ctypes.data, None if grad is None else grad.ctypes.data) != 0:
        raise ValueError('unique_softmax_c: invalid argument')
    return loss, grad


def circle(logits, labels, mask=None, gamma=64.0, margin=0.25, clip=True, want_grad=True):
    """(loss [B], has_pair [B] bool, dlogits [B, L] | None) of CircleLoss (losses_impl.py:1036-1116) on raw logits
    (clip = get_logits' clip_by_value(0, 1)); fp64 inside, the pair sum with its maximum pulled out."""
    logits, labels, m = _prep(logits, labels, mask)
    B, L = logits.shape
    lib = ctypes.CDLL(build()['lw_f64'])
    fn = lib.tfr_c_circle_f64
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_int] + \
        [ctypes.c_void_p] * 3
    loss = np.empty(B, dtype=np.float32)
    has = np.empty(B, dtype=np.uint8)
    grad = np.empty((B, L), dtype=np.float32) if want_grad else None
    if fn(logits.ctypes.data, labels.ctypes.data, None if m is None else m.ctypes.data, B, L, float(gamma), float(margin),
          1 if clip else 0, loss.ctypes.data, has.ctypes.data, None if grad is None else grad.ctypes.data) != 0:
        raise ValueError('circle_c: invalid argument')
    return loss, has.astype(bool), grad
