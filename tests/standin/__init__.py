"""ctypes loader of the native stand-in transcript (tests/standin/standin_transcript.c): test / bench scaffolding."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "standin_transcript.c")
_SO = os.path.join(_HERE, "libstandin_transcript.so")
_lib = None


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-o", _SO, _SRC])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.standin_init.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
        _lib.standin_absorb.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        _lib.standin_squeeze.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        _lib.standin_transcript.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p]
    return _lib


class Transcript:
    """state + the C callback.  `fn(proto)` gives the callback as a function pointer of the caller's prototype (the product's
    and the oracle's CFUNCTYPEs are distinct Python objects of the same C signature); `ctx` is the state pointer to pass along."""

    def __init__(self, seed=1):
        self.state = ctypes.create_string_buffer(48)
        lib().standin_init(self.state, seed)
        self.ctx = ctypes.cast(self.state, ctypes.c_void_p)

    def fn(self, proto):
        return proto(("standin_transcript", lib()))

    def fn_ipa(self, proto):
        """the inner-product argument's round callback (nmx_ipa_transcript_fn / the oracle's) on the same state"""
        return proto(("standin_ipa_transcript", lib()))

    def absorb(self, data):
        b = bytes(data)
        lib().standin_absorb(self.state, b, len(b))

    def squeeze(self):
        out = ctypes.create_string_buffer(32)
        lib().standin_squeeze(self.state, out)
        return out.raw
