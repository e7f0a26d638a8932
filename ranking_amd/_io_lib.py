"""Builds and binds ``libtfr_io.so`` (the host-side input C ABI declared in include/tfr_io.h)."""
from __future__ import annotations

import ctypes
import os
import shutil
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
INCLUDE = os.path.join(os.path.dirname(_HERE), 'include')
LIB_PATH = os.path.join(CSRC, 'libtfr_io.so')
SOURCE = os.path.join(CSRC, 'tfr_io.cpp')
CXX_FLAGS = ['-O2', '-std=c++17', '-fPIC', '-shared', '-pthread', '-fvisibility=default']

_lock = threading.Lock()
_lib = None


class FeatureSpec(ctypes.Structure):
    _fields_ = [('name', ctypes.c_char_p), ('width', ctypes.c_int32), ('default_value', ctypes.c_float)]


_P = ctypes.c_void_p
_SIGNATURES = {
    'tfr_io_abi_version': (ctypes.c_int, []),
    'tfr_io_crc32c': (ctypes.c_uint32, [_P, ctypes.c_size_t]),
    'tfr_io_masked_crc32c': (ctypes.c_uint32, [_P, ctypes.c_size_t]),
    'tfr_io_crc32c_portable': (ctypes.c_uint32, [_P, ctypes.c_size_t]),
    'tfr_io_tfrecord_index': (ctypes.c_int64, [_P, ctypes.c_size_t, ctypes.c_int, _P, _P, ctypes.c_int64]),
    'tfr_io_elwc_max_list_size': (ctypes.c_int64, [_P, _P, ctypes.c_int32]),
    'tfr_io_parse_elwc_batch': (ctypes.c_int, [_P, _P, ctypes.c_int32, ctypes.c_int32, _P, ctypes.c_int32, _P,
                                               ctypes.c_int32, _P, _P, _P, _P, ctypes.c_int32]),
    'tfr_io_parse_elwc_batch_bf16': (ctypes.c_int, [_P, _P, ctypes.c_int32, ctypes.c_int32, _P, ctypes.c_int32, _P,
                                                    ctypes.c_int32, _P, _P, _P, _P, ctypes.c_int32, _P,
                                                    ctypes.c_int32, _P]),
    'tfr_io_f32_to_bf16': (None, [_P, _P, ctypes.c_size_t]),
    'tfr_io_parse_batch': (ctypes.c_int, [ctypes.c_int32, _P, _P, ctypes.c_int32, ctypes.c_int32, _P, ctypes.c_int32, _P,
                                          ctypes.c_int32, _P, _P, _P, _P, _P, ctypes.c_int32, _P, ctypes.c_int32, _P]),
    'tfr_io_max_list_size': (ctypes.c_int64, [ctypes.c_int32, _P, _P, ctypes.c_int32, _P, ctypes.c_int32]),
    'tfr_io_parse_counters': (None, [_P, _P]),
    'tfr_io_libsvm_load': (ctypes.c_int64, [_P, ctypes.c_size_t, ctypes.c_int32, ctypes.c_int32, _P, _P, _P]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)
ABI_VERSION = 2                        # = TFR_IO_ABI_VERSION of include/tfr_io.h

ERRORS = {-1: 'invalid argument', -2: 'truncated or malformed record / protobuf', -3: 'checksum mismatch',
          -4: 'a feature is present with a length different from its spec',
          -5: 'a numeric feature spec matched a bytes_list feature',
          -6: 'an ExampleInExample record without its serialized_context feature'}
FORMAT_ELWC, FORMAT_EIE, FORMAT_SEQ, FORMAT_EXAMPLE = 0, 1, 2, 3


class TfrIoError(RuntimeError):
    pass


def _fingerprint() -> str:
    import hashlib
    h = hashlib.sha256(' '.join(CXX_FLAGS).encode())
    for d in (SOURCE, os.path.join(INCLUDE, 'tfr_io.h')):
        if os.path.exists(d):
            with open(d, 'rb') as f:
                h.update(f.read())
    return h.hexdigest()


def _stale():
    """Content decides (see _lib._stale); mtimes only when there is no stamp."""
    if not os.path.exists(LIB_PATH):
        return True
    try:
        with open(LIB_PATH + '.stamp') as f:
            return f.read().strip() != _fingerprint()
    except OSError:
        t = os.path.getmtime(LIB_PATH)
        return any(os.path.getmtime(d) > t for d in (SOURCE, os.path.join(INCLUDE, 'tfr_io.h')) if os.path.exists(d))


def build(force: bool = False) -> str:
    from ._lib import _BuildLock
    with _lock, _BuildLock(LIB_PATH + '.lock'):              # ranks of one torchrun job build once, not concurrently
        if not force and not _stale():
            return LIB_PATH
        cxx = shutil.which('g++') or shutil.which('c++')
        if cxx is None:
            raise TfrIoError('g++ not found: cannot build %s' % LIB_PATH)
        tmp = '%s.%d.tmp' % (LIB_PATH, os.getpid())
        res = subprocess.run([cxx] + CXX_FLAGS + ['-I', INCLUDE, SOURCE, '-o', tmp], capture_output=True, text=True)
        if res.returncode != 0:
            raise TfrIoError('g++ failed:\n%s\n%s' % (res.stdout, res.stderr))
        os.replace(tmp, LIB_PATH)
        with open(LIB_PATH + '.stamp', 'w') as f:
            f.write(_fingerprint() + '\n')
        return LIB_PATH


def load():
    global _lib
    if _lib is not None:
        return _lib
    if _stale():
        if shutil.which('g++') or shutil.which('c++'):
            build()
        elif not os.path.exists(LIB_PATH):
            raise TfrIoError('%s is missing and no C++ compiler is available' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    got = lib.tfr_io_abi_version()
    if got != ABI_VERSION:
        raise TfrIoError('%s reports ABI version %d, this binding was written against %d (include/tfr_io.h '
                         'TFR_IO_ABI_VERSION): rebuild it' % (LIB_PATH, got, ABI_VERSION))
    _lib = lib
    return lib


def check(code: int, what: str) -> int:
    if code >= 0:
        return code
    msg = '%s: %s (code %d)' % (what, ERRORS.get(int(code), 'error'), code)
    if code in (-1, -4, -5, -6):
        raise ValueError(msg)
    raise TfrIoError(msg)
