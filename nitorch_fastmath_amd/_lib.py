"""ctypes binding of libnfm_hip.so (the C ABI declared in include/nfm_hip.h).

The shared library is the product: there is NO fallback.  If it is missing, or a
tensor does not live on a ROCm device, the call fails loudly.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (kernel experiments load another build of the library: honoured only under NFM_DEBUG, the one variable the
# product reads; the row-wave measurement knobs of nfm_rowwave.hip sit behind the same gate)
LIB_PATH = (os.environ.get('NFM_HIP_LIB') if os.environ.get('NFM_DEBUG') else None) or os.path.join(_HERE, 'libnfm_hip.so')

F32, F64 = 0, 1
MAT_SYM, MAT_DIAG, MAT_SCAL, MAT_FULL = 0, 1, 2, 3
FLAG_TS_PERTURB = 1
EIG_VECTORS, EIG_FAST = 1, 2                 # flags of nfm_qr_eig_sym
MAT_PIVOTED, INVERT_PIVOTED = 16, 2          # include/nfm_hip.h: NFM_MAT_PIVOTED, NFM_INVERT_PIVOTED
RED_NANSUM, RED_NANMAX, RED_NANMIN, RED_SUM, RED_MAX, RED_MIN, RED_NANCOUNT, RED_NANSUMSQ = range(8)
MAX_DIM = 16
SIDE = {'left': 0, 'right': 1, 'both': 2}


class Operand(ctypes.Structure):
    """struct nfm_operand (include/nfm_hip.h)"""
    _fields_ = [('ptr', ctypes.c_void_p),
                ('stride_outer', ctypes.c_int64),
                ('stride_inner', ctypes.c_int64),
                ('stride_row', ctypes.c_int64),
                ('stride_col', ctypes.c_int64)]


_i, _i64, _vp, _op = ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.POINTER(Operand)
_dp = ctypes.POINTER(ctypes.c_double)

# name -> argtypes; every symbol include/nfm_hip.h declares
SIGNATURES = {
    'nfm_sym_solve': [_i, _i, _i, _i64, _i64, _op, _op, _op, _dp, _vp],
    'nfm_sym_matvec': [_i, _i, _i, _i, _i64, _i64, _op, _op, _op, _op, _vp],
    'nfm_sym_invert': [_i, _i, _i, _i64, _i64, _op, _op, _vp],
    'nfm_sym_det': [_i, _i, _i64, _i64, _op, _op, _vp],
    'nfm_sym_to_full': [_i, _i, _i64, _i64, _op, _op, _vp],
    'nfm_sym_outer': [_i, _i, _i64, _i64, _op, _op, _vp],
    'nfm_sym_outer2': [_i, _i, _i, _i64, _i64, _op, _op, _op, _vp],
    'nfm_sym_matmul': [_i, _i, _i, _i, _i64, _i64, _op, _op, _op, _vp],
    'nfm_sym_matmul_solve': [_i, _i, _i, _i, _i64, _i64, _op, _op, _op, _op, _dp, _vp],
    'nfm_batch_inv': [_i, _i, _i, _i64, _i64, _op, _op, _vp],
    'nfm_batch_det': [_i, _i, _i64, _i64, _op, _op, _vp],
    'nfm_batch_matvec': [_i, _i, _i, _i64, _i64, _op, _op, _op, _vp],
    'nfm_reduce_all': [_i, _i, _i, _i64, _vp, _vp, ctypes.c_size_t, _vp, _vp],
    'nfm_reduce_dim_workspace_bytes': [_i, _i, _i64, _i64, _i64, _i],
    'nfm_reduce_dim': [_i, _i, _i, _i64, _i64, _i64, _vp, _vp, ctypes.c_size_t, _vp, _vp, _vp],
    'nfm_reduce_moments_workspace_bytes': [_i, _i64, _i64, _i64],
    'nfm_reduce_moments': [_i, _i64, _i64, _i64, _vp, _vp, ctypes.c_size_t, _vp, _vp],
    'nfm_reduce_stat': [_i, _i, _i, _i64, _i64, _i64, _vp, _vp, ctypes.c_size_t, _vp, _vp],
    'nfm_reduce_median_workspace_bytes': [_i64, _i64],
    'nfm_reduce_median': [_i, _i, _i64, _i64, _vp, _vp, ctypes.c_size_t, _vp, _vp, _vp],
    'nfm_reduce_median_lane_max': [_i],
    'nfm_reduce_median_mid': [_i, _i, _i64, _i64, _i64, _vp, _vp, _vp, _vp],
    'nfm_qr_givens': [_i, _i64, _i64, _op, _op, _vp, _vp],
    'nfm_qr_givens_apply': [_i, _i, _i, _i, _i, _i64, _i64, _op, _op, _op, _vp],
    'nfm_qr_householder': [_i, _i, _i, _i64, _i64, _op, _vp, _vp],
    'nfm_qr_householder_apply': [_i, _i, _i, _i, _i64, _i64, _op, _op, _vp],
    'nfm_qr_hessenberg': [_i, _i, _i, _i, _i, _i64, _i64, _op, _vp, _vp],
    'nfm_qr_qr_hessenberg': [_i, _i, _i64, _i64, _op, _vp, _vp],
    'nfm_qr_rq_hessenberg': [_i, _i, _i, _i64, _i64, _op, _op, _vp, _vp],
    'nfm_qr_eig_sym': [_i, _i, _i, _i, _i, ctypes.c_double, _i64, _i64, _op, _vp, _vp],
}

_lib = None


def lib():
    """Load the HIP backend; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f'nitorch_fastmath_amd: {LIB_PATH} is missing. Build it with '
                '`python -c "import __graft_entry__ as g; g.build()"` or '
                '`make -C nitorch_fastmath_amd/csrc`. There is no CPU fallback.')
        L = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = argtypes
            fn.restype = ctypes.c_size_t if name.endswith('_workspace_bytes') else ctypes.c_int
        L.nfm_strerror.argtypes = [ctypes.c_int]
        L.nfm_strerror.restype = ctypes.c_char_p
        L.nfm_version.argtypes = []
        L.nfm_version.restype = ctypes.c_int
        L.nfm_reduce_workspace_bytes.argtypes = []
        L.nfm_reduce_workspace_bytes.restype = ctypes.c_size_t
        _lib = L
    return _lib


def check(rc):
    """C-ABI status -> Python exception (errors never cross the boundary as C++ throws)."""
    if rc == 0:
        return
    msg = lib().nfm_strerror(rc).decode()
    if rc < 0:
        raise ValueError(f'nitorch_fastmath_amd: {msg} (code {rc})')
    raise RuntimeError(f'nitorch_fastmath_amd: HIP error {rc}: {msg}')
