"""Host-side mirror of the reference's provider interface for the commitment / MSM path.

Same names, argument meaning and error behaviour as the reference (paths relative to /root/reference):
  DlogGroupExt             src/provider/traits.rs:77-117   -> class DlogGroup
  CommitmentEngineTrait    src/traits/commitment.rs:52-195 -> class CommitmentEngine
  Pedersen / HyperKZG CE   src/provider/pedersen.rs:240-305, src/provider/hyperkzg.rs:584-645
Every group operation happens inside libnova_mi355x.so; this module only marshals buffers.

Buffers: scalars are (n, 32) uint8 -- canonical little-endian by default (`to_repr()`), raw Montgomery limbs
with mont=True; small scalars are uint64 arrays; bases are (n, 64) uint8 x||y, identity = 64 zero bytes.
A torch CUDA uint8 / int64 tensor may be passed for scalars: it is used in place (HBM-resident input).
"""
import ctypes
from dataclasses import dataclass

import numpy as np

from . import _lib as L

BN254_G1, GRUMPKIN, PALLAS, VESTA = 0, 1, 2, 3
CURVE_NAMES = {0: "bn254_g1", 1: "grumpkin", 2: "pallas", 3: "vesta"}


class PendingCommitment:
    """ticket of CommitmentEngine.commit_begin"""

    def __init__(self, ticket, partial, keep):
        self.ticket, self.partial, self._keep = ticket, partial, keep

    def finish(self):
        assert self.ticket, "finished already"
        out = _Out(partial=self.partial)
        t, self.ticket = self.ticket, 0
        try:
            _check(L.lib().nmx_commit_finish(t, *out.p))
        finally:
            self._keep = None
        return out.get()


class NmxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"nmx error {code}: {msg}")
        self.code = code


def _check(rc):
    if rc != 0:
        raise NmxError(rc, L.lib().nmx_last_error().decode())


def init_devices(count=0, oversubscribe=False):
    """nmx_init_devices: keys registered afterwards are sharded over `count` devices of THIS process (0 = all visible);
    oversubscribe maps logical devices onto fewer GPUs (tests).  Returns the number of devices in use."""
    _check(L.lib().nmx_init_devices(count, L.DEVICES_OVERSUBSCRIBE if oversubscribe else 0))
    return L.lib().nmx_devices_in_use()


def shard_plan(n_key, k, offset, n):
    """[(device, offset inside the shard, count)] for a call over key[offset, offset + n) of a k-way sharded key."""
    buf = (ctypes.c_size_t * (3 * max(k, 1)))()
    cnt = L.lib().nmx_shard_plan(n_key, k, offset, n, buf, max(k, 1))
    if cnt < 0:
        raise NmxError(cnt, "bad shard_plan arguments")
    return [(buf[3 * i], buf[3 * i + 1], buf[3 * i + 2]) for i in range(cnt)]


class ShardedVector:
    """A field vector resident shard by shard in the HBM of the devices that hold the matching points of a key of `n_key`
    points (nmx_svec_*): the layout the reference's own decomposition uses -- coefficients and bases chunked together,
    /root/reference/src/provider/msm.rs:564-574.  Elements are raw 32-byte words."""

    def __init__(self, n_key, n):
        h = ctypes.c_uint64(0)
        _check(L.lib().nmx_svec_alloc(n_key, n, ctypes.byref(h)))
        self.handle, self.n_key, self.n = h.value, n_key, n

    @classmethod
    def for_key(cls, ck, elems):
        """Laid out like the registered key `ck` (its registered length, not len(ck): see CommitmentKey.shard_plan)."""
        return cls.from_host(ck.registered_len(), elems)

    @classmethod
    def from_host(cls, n_key, elems):
        a = _host_u8(elems, 32)
        v = cls(n_key, a.size // 32)
        _check(L.lib().nmx_svec_write(v.handle, a.ctypes.data))
        return v

    def parts(self):
        """[(device pointer, element count, HIP device ordinal)] in shard order."""
        cap = 64
        ptrs, cnts, devs = (ctypes.c_void_p * cap)(), (ctypes.c_size_t * cap)(), (ctypes.c_int * cap)()
        k = L.lib().nmx_svec_parts(self.handle, ptrs, cnts, devs, cap)
        if k < 0:
            raise NmxError(k, L.lib().nmx_last_error().decode())
        return [(ptrs[i] or 0, cnts[i], devs[i]) for i in range(k)]

    def to_host(self):
        out = np.zeros((self.n, 32), dtype=np.uint8)
        _check(L.lib().nmx_svec_read(self.handle, out.ctypes.data))
        return out

    def __len__(self):
        return self.n

    def close(self):
        if self.handle:
            _check(L.lib().nmx_svec_free(self.handle))
            self.handle = 0


def svec_map(field, op, ins, challenge=None, out=None, mont=False):
    """nmx_svec_map: an element-wise NIFS kernel (L.OP_*) over sharded vectors, every device on its own piece."""
    if out is None:
        out = ShardedVector(ins[0].n_key, ins[0].n)
    hs = (ctypes.c_uint64 * len(ins))(*[v.handle for v in ins])
    ch = None if challenge is None else _host_u8(challenge, 32)
    _check(L.lib().nmx_svec_map(field, op, hs, len(ins), None if ch is None else ch.ctypes.data,
                                L.SCALARS_MONT if mont else 0, out.handle))
    return out


def _is_sharded_list(x):
    return isinstance(x, (list, tuple)) and len(x) > 0 and all(_is_device_tensor(t) for t in x)


def _is_device_tensor(x):
    return hasattr(x, "data_ptr") and getattr(x, "is_cuda", False)


def _host_u8(x, width):
    """-> contiguous uint8 array with trailing dimension `width` (or empty)."""
    if isinstance(x, (bytes, bytearray, memoryview)):
        x = np.frombuffer(bytes(x), dtype=np.uint8)
    a = np.ascontiguousarray(x, dtype=np.uint8).reshape(-1)
    assert a.size % width == 0, f"buffer length {a.size} is not a multiple of {width}"
    return a


def _scalar_arg(s, width):
    """-> (pointer, n, device_flag, keepalive)"""
    if _is_device_tensor(s):
        assert s.is_contiguous()
        nbytes = s.numel() * s.element_size()
        assert nbytes % width == 0
        return s.data_ptr(), nbytes // width, L.SCALARS_DEVICE, s
    if width == 8:
        a = np.ascontiguousarray(s, dtype=np.uint64).reshape(-1)
        return a.ctypes.data, a.size, 0, a
    a = _host_u8(s, width)
    return a.ctypes.data, a.size // width, 0, a


@dataclass(frozen=True)
class Commitment:
    """Affine result, as `to_coordinates()` returns it (src/provider/traits.rs:303-312)."""
    xy: bytes          # canonical x||y, 64 bytes; zeros for the identity
    is_inf: bool

    def to_coordinates(self):
        return (int.from_bytes(self.xy[:32], "little"), int.from_bytes(self.xy[32:], "little"), self.is_inf)


class _Out:
    def __init__(self, k=1, partial=False):
        self.w = 128 if partial else 64
        self.buf = np.zeros(self.w * max(k, 1), dtype=np.uint8)
        self.inf = np.zeros(max(k, 1), dtype=np.uint8)

    @property
    def p(self):
        return self.buf.ctypes.data, self.inf.ctypes.data

    def get(self, j=0):
        return Commitment(self.buf[self.w * j: self.w * (j + 1)].tobytes(), bool(self.inf[j]))


class CommitmentKey:
    """A commitment key resident in HBM: `ck` bases + blinding generator `h`
    (src/provider/pedersen.rs:33-45, src/provider/hyperkzg.rs:84-100).  The reference passes `&ck.ck[..n]`
    slices; here the key is registered once (nmx_bases_register) and calls address a prefix of it."""

    def __init__(self, curve, handle, n, h_xy64, mont=False):
        self.curve, self.handle, self.n, self.mont = curve, handle, n, mont
        self.h = bytes(h_xy64)

    @classmethod
    def from_host(cls, curve, ck_xy64, h_xy64=None, mont=False, precompute=True):
        """Register a host key.  precompute=True also builds the window tables 2^(cw) * P_i in HBM (16x the key for
        c = 16): commitment keys are long-lived, so every later MSM runs all windows into one bucket set."""
        a = _host_u8(ck_xy64, 64)
        n = a.size // 64
        h = ctypes.c_uint64(0)
        flags = (L.BASES_MONT if mont else 0) | (L.BASES_PRECOMPUTE if precompute else 0)
        _check(L.lib().nmx_bases_register(curve, a.ctypes.data, n, flags, ctypes.byref(h)))
        return cls(curve, h.value, n, h_xy64 if h_xy64 is not None else bytes(64), mont)

    @classmethod
    def from_host_validated(cls, curve, ck_xy64, h_xy64=None, mont=False, precompute=True):
        """from_host + the checks `read_points` applies to loaded keys (ptau.rs:372-391): NmxError(E_POINT) on a
        non-canonical coordinate or an off-curve point."""
        a = _host_u8(ck_xy64, 64)
        h = ctypes.c_uint64(0)
        flags = (L.BASES_MONT if mont else 0) | (L.BASES_PRECOMPUTE if precompute else 0) | L.BASES_VALIDATE
        _check(L.lib().nmx_bases_register(curve, a.ctypes.data, a.size // 64, flags, ctypes.byref(h)))
        return cls(curve, h.value, a.size // 64, h_xy64 if h_xy64 is not None else bytes(64), mont)

    @classmethod
    def load_ptau(cls, curve, path, n, h_xy64=None, precompute=True):
        """HyperKZG `load_setup` (src/provider/hyperkzg.rs:658-674): ck = the first n.next_power_of_two() tauG1 points of
        a .ptau file, streamed to HBM.  `h` (from_label on the reference side) and tau_H (G2) stay with the host."""
        num = 1 if n <= 1 else 1 << (n - 1).bit_length()
        h = ctypes.c_uint64(0)
        _check(L.lib().nmx_bases_register_ptau(curve, str(path).encode(), num, 2, L.BASES_PRECOMPUTE if precompute else 0,
                                               ctypes.byref(h)))
        return cls(curve, h.value, num, h_xy64 if h_xy64 is not None else bytes(64))

    @classmethod
    def load_keyfile(cls, curve, path, n, precompute=True):
        """Pedersen `load_setup` (src/provider/pedersen.rs:318-340): "PEDERSEN_KEY" | h | ck[0..n.next_power_of_two())."""
        num = 1 if n <= 1 else 1 << (n - 1).bit_length()
        h = ctypes.c_uint64(0)
        hxy = np.zeros(64, dtype=np.uint8)
        _check(L.lib().nmx_bases_register_keyfile(curve, str(path).encode(), num, L.BASES_PRECOMPUTE if precompute else 0,
                                                  ctypes.byref(h), hxy.ctypes.data))
        return cls(curve, h.value, num, hxy.tobytes())

    @classmethod
    def generate(cls, curve, n, k0=1, precompute=True):
        """Synthetic key P_i = (k0 + i) * G built on the device (nmx_bases_generate); h = P_n."""
        h = ctypes.c_uint64(0)
        _check(L.lib().nmx_bases_generate(curve, k0, n + 1, L.BASES_PRECOMPUTE if precompute else 0, ctypes.byref(h)))
        key = cls(curve, h.value, n, bytes(64))
        key.h = key.read(n, 1).tobytes()
        return key

    def shard_plan(self, offset=0, n=None):
        """[(device, offset inside the shard, count)] for a call over key[offset, offset + n), from the layout the
        REGISTERED key has (nmx_bases_shard_plan) -- the plan to cut shard-resident scalar pieces by.  (A generated key is
        registered with its blinding point h behind it: n + 1 points, so shard_plan(len(key), ...) would cut elsewhere.)"""
        n = self.n - offset if n is None else n
        buf = (ctypes.c_size_t * (3 * 64))()
        cnt = L.lib().nmx_bases_shard_plan(self.handle, offset, n, buf, 64, None)
        if cnt < 0:
            raise NmxError(cnt, L.lib().nmx_last_error().decode())
        return [(buf[3 * i], buf[3 * i + 1], buf[3 * i + 2]) for i in range(cnt)]

    def registered_len(self):
        """Number of points registered under the handle (len(self) + 1 for a generated key: h sits behind ck)."""
        nk = ctypes.c_size_t(0)
        cnt = L.lib().nmx_bases_shard_plan(self.handle, 0, 0, None, 0, ctypes.byref(nk))
        if cnt < 0:
            raise NmxError(cnt, L.lib().nmx_last_error().decode())
        return nk.value

    def read(self, offset, n):
        out = np.zeros((n, 64), dtype=np.uint8)
        _check(L.lib().nmx_bases_read(self.handle, offset, n, out.ctypes.data))
        return out

    def __len__(self):
        return self.n

    def close(self):
        if self.handle:
            _check(L.lib().nmx_bases_unregister(self.handle))
            self.handle = 0


class DlogGroup:
    """`DlogGroupExt` for one curve (src/provider/traits.rs:77-117)."""

    def __init__(self, curve):
        assert curve in CURVE_NAMES
        self.curve = curve

    def min_gpu_n(self):
        """Smallest n the reference-side shim should send to the GPU (below it the CPU `msm()` wins: IPA's two-point
        MSMs, src/provider/pedersen.rs:484-497; msm_simple for n <= 16, src/provider/msm.rs:233).  This mirror has no
        CPU path, so it only reports the rule."""
        return int(L.lib().nmx_min_gpu_n(self.curve))

    # -- vartime_multiscalar_mul (traits.rs:79; msm(), src/provider/msm.rs:225) -------------------------
    def vartime_multiscalar_mul(self, scalars, bases, mont=False, partial=False, offset=0, nocache=False):
        """offset: with a CommitmentKey, use bases[offset .. offset + n) of it (`&ck.ck[offset..][..n]`).
        With a host array (the trait's slice form) the library's slice cache makes the array resident on first
        sight -- pass the SAME array object (or a prefix view of it) again and no bases move; nocache=True uploads
        for this call only.
        Shard-resident scalars over a multi-device key: a ShardedVector, or a list of CUDA tensors -- one per piece of
        shard_plan(len(key), devices, offset, n), each on that piece's device (NMX_SCALARS_SHARDED)."""
        if isinstance(scalars, ShardedVector):
            assert isinstance(bases, CommitmentKey) and offset == 0
            out = _Out(partial=partial)
            flags = (L.SCALARS_MONT if mont else 0) | (L.OUT_PARTIAL if partial else 0)
            _check(L.lib().nmx_msm_svec(bases.handle, scalars.handle, len(scalars), flags, *out.p))
            return out.get()
        if _is_sharded_list(scalars):
            assert isinstance(bases, CommitmentKey)
            ptrs = (ctypes.c_void_p * len(scalars))(*[t.data_ptr() for t in scalars])
            n = sum(t.numel() * t.element_size() for t in scalars) // 32
            assert offset + n <= bases.n
            out = _Out(partial=partial)
            flags = L.SCALARS_SHARDED | (L.SCALARS_MONT if mont else 0) | (L.OUT_PARTIAL if partial else 0)
            _check(L.lib().nmx_msm_handle(bases.handle, offset, ptrs, n, flags, *out.p))
            return out.get()
        sp, n, dev, _k = _scalar_arg(scalars, 32)
        flags = dev | (L.SCALARS_MONT if mont else 0) | (L.OUT_PARTIAL if partial else 0) | (L.BASES_NOCACHE if nocache else 0)
        out = _Out(partial=partial)
        if isinstance(bases, CommitmentKey):
            assert bases.curve == self.curve
            assert offset + n <= bases.n, "assert_eq!(coeffs.len(), bases.len()) / ck.len() >= v.len()"
            _check(L.lib().nmx_msm_handle(bases.handle, offset, sp, n, flags, *out.p))
        else:
            b = _host_u8(bases, 64)
            assert b.size // 64 == n, "assert_eq!(coeffs.len(), bases.len())  (msm.rs:226)"
            _check(L.lib().nmx_msm(self.curve, sp, b.ctypes.data, n, flags | (L.BASES_MONT if mont else 0), *out.p))
        return out.get()

    # -- batch_vartime_multiscalar_mul (traits.rs:82-90; blitzar.rs:22-40) -----------------------------
    def batch_vartime_multiscalar_mul(self, scalar_vecs, bases, mont=False):
        k = len(scalar_vecs)
        args = [_scalar_arg(v, 32) for v in scalar_vecs]
        dev = {a[2] for a in args}
        assert len(dev) <= 1, "all vectors must live on the same side"
        flags = (dev.pop() if dev else 0) | (L.SCALARS_MONT if mont else 0)
        ptrs = (ctypes.c_void_p * max(k, 1))(*[a[0] for a in args])
        lens = (ctypes.c_size_t * max(k, 1))(*[a[1] for a in args])
        out = _Out(k)
        if isinstance(bases, CommitmentKey):
            _check(L.lib().nmx_msm_batch_handle(bases.handle, ptrs, lens, k, flags, *out.p))
        else:
            b = _host_u8(bases, 64)
            _check(L.lib().nmx_msm_batch(self.curve, ptrs, lens, k, b.ctypes.data, b.size // 64,
                                         flags | (L.BASES_MONT if mont else 0), *out.p))
        return [out.get(j) for j in range(k)]

    # -- batch_vartime_multiscalar_mul_small (traits.rs:109-117) --------------------------------------------
    def batch_vartime_multiscalar_mul_small(self, scalar_vecs, bases, max_num_bits=None):
        k = len(scalar_vecs)
        args = [_scalar_arg(v, 8) for v in scalar_vecs]
        dev = {a[2] for a in args}
        assert len(dev) <= 1, "all vectors must live on the same side"
        flags = dev.pop() if dev else 0
        bits = L.BITS_AUTO if max_num_bits is None else int(max_num_bits)
        ptrs = (ctypes.c_void_p * max(k, 1))(*[a[0] for a in args])
        lens = (ctypes.c_size_t * max(k, 1))(*[a[1] for a in args])
        out = _Out(k)
        if isinstance(bases, CommitmentKey):
            _check(L.lib().nmx_msm_u64_batch_handle(bases.handle, ptrs, lens, k, bits, flags, *out.p))
        else:
            b = _host_u8(bases, 64)
            _check(L.lib().nmx_msm_u64_batch(self.curve, ptrs, lens, k, b.ctypes.data, b.size // 64, bits, flags, *out.p))
        return [out.get(j) for j in range(k)]

    # -- vartime_multiscalar_mul_small* (traits.rs:93-106; msm.rs:469-503) ---------------------------------
    def vartime_multiscalar_mul_small(self, scalars, bases, partial=False):
        return self.vartime_multiscalar_mul_small_with_max_num_bits(scalars, bases, None, partial)

    def vartime_multiscalar_mul_small_with_max_num_bits(self, scalars, bases, max_num_bits, partial=False):
        sp, n, dev, _k = _scalar_arg(scalars, 8)
        bits = L.BITS_AUTO if max_num_bits is None else int(max_num_bits)
        flags = dev | (L.OUT_PARTIAL if partial else 0)
        out = _Out(partial=partial)
        if isinstance(bases, CommitmentKey):
            assert n <= bases.n
            _check(L.lib().nmx_msm_u64_handle(bases.handle, 0, sp, n, bits, flags, *out.p))
        else:
            b = _host_u8(bases, 64)
            assert b.size // 64 == n, "assert_eq!(bases.len(), scalars.len())  (msm.rs:486)"
            _check(L.lib().nmx_msm_u64(self.curve, sp, b.ctypes.data, n, bits, flags, *out.p))
        return out.get()

    def point_sum(self, partials):
        """Sum 128-byte partials (sharded MSM combine)."""
        buf = np.frombuffer(b"".join(partials), dtype=np.uint8) if not isinstance(partials, np.ndarray) else partials
        buf = np.ascontiguousarray(buf, dtype=np.uint8).reshape(-1)
        out = _Out()
        _check(L.lib().nmx_point_sum(self.curve, buf.ctypes.data, buf.size // 128, *out.p))
        return out.get()


class CommitmentEngine:
    """`CommitmentEngineTrait` restricted to the hot path (src/traits/commitment.rs:52-195)."""

    def __init__(self, curve):
        self.group = DlogGroup(curve)

    def setup_synthetic(self, n, k0=1):
        return CommitmentKey.generate(self.group.curve, n, k0)

    def commit(self, ck, v, r=None, mont=False, partial=False):
        """msm(v, ck[..len v]) + h*r  (pedersen.rs:263-270, hyperkzg.rs:584-591).  v may be a ShardedVector."""
        if isinstance(v, ShardedVector):
            rr = _host_u8(bytes(32) if r is None else r, 32)
            hh = _host_u8(ck.h, 64)
            flags = (L.SCALARS_MONT if mont else 0) | (L.BASES_MONT if ck.mont else 0) | (L.OUT_PARTIAL if partial else 0)
            out = _Out(partial=partial)
            _check(L.lib().nmx_commit_svec(ck.handle, v.handle, len(v), hh.ctypes.data, rr.ctypes.data, flags, *out.p))
            return out.get()
        sp, n, dev, _k = _scalar_arg(v, 32)
        assert len(ck) >= n, "assert!(ck.ck.len() >= v.len())"
        rr = _host_u8(bytes(32) if r is None else r, 32)
        hh = _host_u8(ck.h, 64)
        flags = dev | (L.SCALARS_MONT if mont else 0) | (L.BASES_MONT if ck.mont else 0) | (L.OUT_PARTIAL if partial else 0)
        out = _Out(partial=partial)
        _check(L.lib().nmx_commit(ck.handle, sp, n, hh.ctypes.data, rr.ctypes.data, flags, *out.p))
        return out.get()

    def commit_begin(self, ck, v, r=None, mont=False, partial=False):
        """nmx_commit_begin: the commitment runs beside the caller's next calls (commit(W) beside cross term + commit(T):
        r1cs/mod.rs:590-622 never reads comm_W); .finish() returns what commit() would have.  v must stay alive and
        unchanged until then (the object keeps a reference)."""
        sp, n, dev, keep = _scalar_arg(v, 32)
        assert len(ck) >= n, "assert!(ck.ck.len() >= v.len())"
        rr = _host_u8(bytes(32) if r is None else r, 32)
        hh = _host_u8(ck.h, 64)
        flags = dev | (L.SCALARS_MONT if mont else 0) | (L.BASES_MONT if ck.mont else 0) | (L.OUT_PARTIAL if partial else 0)
        t = ctypes.c_uint64(0)
        _check(L.lib().nmx_commit_begin(ck.handle, sp, n, hh.ctypes.data, rr.ctypes.data, flags, ctypes.byref(t)))
        return PendingCommitment(t.value, partial, (v, keep, ck))

    def batch_commit(self, ck, vs, rs=None, mont=False):
        """hyperkzg.rs:593-612: batch MSM over ck[..max len], then + h*r_i each."""
        rs = [None] * len(vs) if rs is None else rs
        assert len(vs) == len(rs)
        if all(r is None or bytes(r) == bytes(32) for r in rs):
            return self.group.batch_vartime_multiscalar_mul(vs, ck, mont)
        return [self.commit(ck, v, r, mont) for v, r in zip(vs, rs)]

    def _blind(self, ck, point, r, mont):
        """point + h*r through the ABI: both as partials, summed by nmx_point_sum."""
        if r is None or bytes(_host_u8(r, 32)) == bytes(32):
            return self.group.point_sum([point.xy])
        blind = self.commit(ck, np.zeros((0, 32), np.uint8), r, mont, partial=True)
        return self.group.point_sum([point.xy, blind.xy])

    def _sparse(self, ck, indices, scalars, mont):
        idx = np.ascontiguousarray(indices, dtype=np.uint64)
        out = _Out(partial=True)
        if scalars is None:
            sp, flags = None, 0
        else:
            sp, n, dev, _k = _scalar_arg(scalars, 32)
            assert n == len(idx), "assert_eq!(indices.len(), scalars.len())  (pedersen.rs:416)"
            flags = dev | (L.SCALARS_MONT if mont else 0)
        _check(L.lib().nmx_msm_sparse_handle(ck.handle, idx.ctypes.data, sp, len(idx), flags | L.OUT_PARTIAL, *out.p))
        return out.get()

    def commit_sparse_binary(self, ck, non_zero_indices, r=None, mont=False):
        """batch_add(&ck.ck, indices) + h*r  (pedersen.rs:395-408, msm.rs:689-708)."""
        return self._blind(ck, self._sparse(ck, non_zero_indices, None, mont), r, mont)

    def commit_sparse(self, ck, indices, scalars, r=None, mont=False):
        """msm(scalars, ck[indices]) + h*r  (pedersen.rs:410-427)."""
        return self._blind(ck, self._sparse(ck, indices, scalars, mont), r, mont)

    def commit_small_range(self, ck, v_u64, r, start, stop, max_num_bits, mont=False):
        """msm_small_with_max_num_bits(v[range], ck[range]) + h*r  (pedersen.rs:285-305)."""
        v = np.ascontiguousarray(v_u64, dtype=np.uint64)[start:stop]
        out = _Out(partial=True)
        assert stop <= len(ck), "assert!(bases.len() == scalars.len())"
        _check(L.lib().nmx_msm_u64_handle(ck.handle, start, v.ctypes.data, len(v), int(max_num_bits), L.OUT_PARTIAL, *out.p))
        return self._blind(ck, out.get(), r, mont)

    def batch_commit_small(self, ck, vs, rs=None, mont=False):
        """commitment.rs:139-150 / hyperkzg.rs:626-645: commit_small per vector."""
        rs = [None] * len(vs) if rs is None else rs
        assert len(vs) == len(rs)
        if all(r is None for r in rs):   # no blinding terms: one batch call (nmx_msm_u64_batch_handle)
            return self.group.batch_vartime_multiscalar_mul_small(vs, ck)
        return [self.commit_small(ck, v, r, mont) for v, r in zip(vs, rs)]

    def commit_small(self, ck, v_u64, r=None, mont=False):
        """msm_small(v, ck[..len v]) + h*r  (pedersen.rs:272-283)."""
        small = self.group.vartime_multiscalar_mul_small(v_u64, ck, partial=True)
        blind = self.commit(ck, np.zeros((0, 32), np.uint8), r, mont, partial=True)
        return self.group.point_sum([small.xy, blind.xy])


def as_ipa_transcript(fn):
    """fn(L_xy64: bytes, L_is_identity: bool, R_xy64: bytes, R_is_identity: bool) -> 32-byte challenge, wrapped as
    nmx_ipa_transcript_fn.  The transcript is the caller's: `absorb(b"L", &L); absorb(b"R", &R); squeeze(b"r")`
    (src/provider/ipa_pc.rs:231-234)."""
    if isinstance(fn, L.IPA_TRANSCRIPT_FN):
        return fn

    def cb(_ctx, Lp, Li, Rp, Ri, out):
        try:
            ch = fn(bytes(Lp[:64]), bool(Li), bytes(Rp[:64]), bool(Ri))
            ctypes.memmove(out, ch, 32)
            return 0
        except Exception:                   # an exception must not cross the C frame: the call fails with NMX_E_ARG
            import traceback
            traceback.print_exc()
            return 1
    return L.IPA_TRANSCRIPT_FN(cb)


def ipa_prove(ck, ck_c_xy64, a, b, transcript, mont=False, ctx=None):
    """InnerProductArgument::prove (src/provider/ipa_pc.rs:174-281) over a registered Pedersen key: `ck` a CommitmentKey (its first
    len(a) points are used), `ck_c_xy64` the already scaled one-point key `ck_c.scale(&r)`, a / b the witness and the instance's
    vector (host arrays or CUDA tensors, len a power of two).  Returns (L_vec, R_vec as 64-byte strings, [(L is identity, R is
    identity)], a_hat as 32 bytes).  The key is never folded (nova_amd/csrc/ipa.hpp)."""
    pa, n, deva, _ka = _scalar_arg(a, 32)
    pb, nb, devb, _kb = _scalar_arg(b, 32)
    assert n == nb and deva == devb, "InvalidInputLength (ipa_pc.rs:185-187) / both vectors on the same side"
    rounds = max(n.bit_length() - 1, 0)
    u = _host_u8(ck_c_xy64, 64)
    oL, oR = np.zeros(64 * max(rounds, 1), np.uint8), np.zeros(64 * max(rounds, 1), np.uint8)
    oi, ah = np.zeros(2 * max(rounds, 1), np.uint8), np.zeros(32, np.uint8)
    flags = deva | (L.SCALARS_MONT if mont else 0) | (L.BASES_MONT if ck.mont else 0)
    cb = as_ipa_transcript(transcript)
    _check(L.lib().nmx_ipa_prove(ck.handle, u.ctypes.data, pa, pb, n, flags, cb, ctx, oL.ctypes.data, oR.ctypes.data, oi.ctypes.data,
                                 ah.ctypes.data))
    Lb, Rb = oL.tobytes(), oR.tobytes()
    return ([Lb[64 * j: 64 * j + 64] for j in range(rounds)], [Rb[64 * j: 64 * j + 64] for j in range(rounds)],
            [(bool(oi[2 * j]), bool(oi[2 * j + 1])) for j in range(rounds)], ah.tobytes())
