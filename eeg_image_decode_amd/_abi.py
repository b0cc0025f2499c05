"""ctypes mirror of include/eegclip.h (struct layouts + prototypes).  Pure declarations, no compute."""
import ctypes as C

BIG = 1 << 62


class Dim(C.Structure):
    _fields_ = [("div", C.c_longlong), ("so", C.c_longlong), ("si", C.c_longlong)]


def dim(si, div=BIG, so=0):
    """offset(i) = (i // div) * so + (i % div) * si   (plain stride when div is omitted)."""
    return Dim(int(div), int(so), int(si))


class GemmDesc(C.Structure):
    _fields_ = [
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
        ("A", C.c_void_p), ("Am", Dim), ("Ak", Dim),
        ("B", C.c_void_p), ("Bk", Dim), ("Bn", Dim),
        ("C", C.c_void_p), ("Cm", Dim), ("Cn", Dim),
        ("Cpre", C.c_void_p),
        ("bias_n", C.c_void_p), ("bias_m", C.c_void_p),
        ("R", C.c_void_p), ("Rm", Dim), ("Rn", Dim),
        ("alpha", C.c_float), ("accumulate", C.c_int), ("act", C.c_int),
        ("drop_p", C.c_float), ("seed", C.c_ulonglong), ("drop_site", C.c_uint),
        ("split_k", C.c_int),
    ]


ACT_NONE, ACT_GELU = 0, 1
_P, _I, _F, _L, _U64 = C.c_void_p, C.c_int, C.c_float, C.c_longlong, C.c_ulonglong

# name -> argtypes   (restype is always int)
PROTOTYPES = {
    "eegclip_abi_version": [],
    "eegclip_gemm_f32": [C.POINTER(GemmDesc), _P],
}


def declare(lib):
    """Attach prototypes; raises AttributeError if the library lacks a symbol the header declares."""
    for name, args in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype = C.c_int
        fn.argtypes = args
    return lib
