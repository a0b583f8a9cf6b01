"""TEST TOOLING, NOT PyTorch: the few names of the torch API that envidr_amd/_lib.py and envidr_amd/fused.py touch on the way to the C ABI --
device memory from hipMalloc of the ROCm installation's HIP runtime (/opt/rocm/lib/libamdhip64.so; this image has no /opt/rocm/lib/asan), numpy on the host side.
PyTorch itself cannot run under the device sanitizer here: the wheel carries its own libamdhip64 / libhsa-runtime64, and the sanitizer
runtime's interceptor of hsa_amd_memory_pool_allocate fails inside them (profiles/r06a/asan_torch_attempt.txt).  tools/asan_driver.py puts
this directory in front of sys.path; nothing else ever imports it."""
import ctypes
import os

import numpy as np

DRY = os.environ.get("ENVIDR_ASAN_SHIM_DRY") == "1"      # no GPU: "device" blocks are host memory, so the stand-in itself can be exercised


class _HostHip:
    """ENVIDR_ASAN_SHIM_DRY=1: the four runtime calls on host memory (the library's launches then fail with 'no device', which the driver
    counts apart from failures of this stand-in)"""
    _libc = ctypes.CDLL(None)
    _libc.aligned_alloc.restype = ctypes.c_void_p
    _libc.aligned_alloc.argtypes = [ctypes.c_size_t, ctypes.c_size_t]
    _libc.free.argtypes = [ctypes.c_void_p]

    def hipMalloc(self, pp, n):
        pp._obj.value = self._libc.aligned_alloc(256, (n + 255) // 256 * 256)      # (hipMalloc blocks are at least so aligned)
        return 0

    def hipFree(self, p):
        self._libc.free(p)
        return 0

    def hipMemcpy(self, dst, src, n, kind):
        ctypes.memmove(dst, src, n)
        return 0

    def hipMemset(self, p, v, n):
        ctypes.memset(p, v, n)
        return 0

    def hipDeviceSynchronize(self):
        return 0


if DRY:
    _hip = _HostHip()
else:
    _hip = ctypes.CDLL(os.environ.get("ENVIDR_ASAN_HIP", "/opt/rocm/lib/libamdhip64.so"), mode=ctypes.RTLD_GLOBAL)
if not DRY:
    _hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    _hip.hipFree.argtypes = [ctypes.c_void_p]
    _hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    _hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed with hipError {rc}")


class _DType:
    def __init__(self, name, np_dtype):
        self.name, self.np = name, np.dtype(np_dtype)

    def __repr__(self):
        return "torch." + self.name


float32, float16, int32, int16, int64, uint8 = (_DType("float32", np.float32), _DType("float16", np.float16), _DType("int32", np.int32),
                                                _DType("int16", np.int16), _DType("int64", np.int64), _DType("uint8", np.uint8))
_bool = _DType("bool", np.bool_)
half = float16
_BY_NP = {d.np: d for d in (float32, float16, int32, int16, int64, uint8, _bool)}


class _Device:
    def __init__(self, kind="cuda", index=None):
        if isinstance(kind, _Device):
            kind, index = kind.type, kind.index
        if isinstance(kind, str) and ":" in kind:
            kind, index = kind.split(":")[0], int(kind.split(":")[1])
        self.type, self.index = kind, (0 if index is None and kind == "cuda" else index)

    def __eq__(self, o):
        return isinstance(o, _Device) and (self.type, self.index) == (o.type, o.index)

    def __hash__(self):
        return hash((self.type, self.index))

    def __repr__(self):
        return f"device({self.type!r}, {self.index})"


class Tensor:
    """a contiguous array: on the host (numpy) or on the GPU (a hipMalloc block)"""

    def __init__(self, shape, dt, host=None, ptr=None, owner=None):
        self.shape, self.dtype = tuple(int(s) for s in shape), dt
        self._host, self._ptr, self._owner = host, ptr, owner
        self.device = _Device("cpu") if host is not None else _Device("cuda", 0)

    @property
    def is_cuda(self):
        return self._host is None

    def numel(self):
        return int(np.prod(self.shape)) if self.shape else 1

    def dim(self):
        return len(self.shape)

    def element_size(self):
        return self.dtype.np.itemsize

    def stride(self, i=None):
        st, acc = [], 1
        for s in reversed(self.shape):
            st.append(acc)
            acc *= s
        st = tuple(reversed(st))
        return st if i is None else st[i]

    def is_contiguous(self):
        return True

    def contiguous(self):
        return self

    def detach(self):
        return self

    def data_ptr(self):
        return self._host.ctypes.data if self._host is not None else (self._ptr or 0)

    def _view(self, shape):
        shape = list(shape[0]) if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else list(shape)
        if -1 in shape:
            known = int(np.prod([s for s in shape if s != -1])) or 1
            shape[shape.index(-1)] = self.numel() // known
        assert int(np.prod(shape)) == self.numel(), (shape, self.shape)
        return Tensor(shape, self.dtype, host=None if self._host is None else self._host.reshape(shape), ptr=self._ptr, owner=self)

    view = reshape = lambda self, *shape: self._view(shape)

    def float(self):
        return self.to(dtype=float32)

    def to(self, dev=None, dtype=None, device=None):
        target = dev if dev is not None else device
        if isinstance(target, _DType):
            target, dtype = None, target
        out = self
        if dtype is not None and dtype is not out.dtype:
            out = from_numpy(out.cpu().numpy().astype(dtype.np))
            if self.is_cuda and target is None:
                target = self.device
        if target is not None and _Device(target).type == "cuda" and not out.is_cuda:
            nbytes = max(out._host.nbytes, 4)
            p = ctypes.c_void_p()
            _check(_hip.hipMalloc(ctypes.byref(p), nbytes), "hipMalloc")
            src = np.ascontiguousarray(out._host)
            if src.nbytes:
                _check(_hip.hipMemcpy(p, src.ctypes.data, src.nbytes, 1), "hipMemcpy H2D")
            out = Tensor(out.shape, out.dtype, ptr=p.value, owner=_Block(p.value))
        elif target is not None and _Device(target).type == "cpu" and out.is_cuda:
            out = out.cpu()
        return out

    def cuda(self):
        return self.to(_Device("cuda", 0))

    def cpu(self):
        if not self.is_cuda:
            return self
        host = np.empty(self.shape, self.dtype.np)
        _check(_hip.hipDeviceSynchronize(), "hipDeviceSynchronize")
        if host.nbytes:
            _check(_hip.hipMemcpy(host.ctypes.data, self._ptr, host.nbytes, 2), "hipMemcpy D2H")
        return Tensor(self.shape, self.dtype, host=host)

    def numpy(self):
        assert not self.is_cuda
        return self._host

    def clone(self):
        return from_numpy(self.cpu().numpy().copy()).to(self.device)

    def zero_(self):
        if self.is_cuda:
            _check(_hip.hipMemset(self._ptr, 0, self.numel() * self.element_size()), "hipMemset")
        else:
            self._host[...] = 0
        return self

    def item(self):
        return self.cpu().numpy().reshape(-1)[0].item()

    def __len__(self):
        return self.shape[0]

    def __int__(self):
        return int(self.item())

    def __iter__(self):
        return (self[i] for i in range(self.shape[0]))

    def _arith(self, other, fn):
        """elementwise arithmetic the slow way (through the host): only the few host-side expressions of fused.py use it"""
        b = other.cpu().numpy() if isinstance(other, Tensor) else other
        return from_numpy(np.asarray(fn(self.cpu().numpy(), b)).astype(self.dtype.np)).to(self.device)

    __add__ = __radd__ = lambda self, o: self._arith(o, lambda a, b: a + b)
    __mul__ = __rmul__ = lambda self, o: self._arith(o, lambda a, b: a * b)
    __sub__ = lambda self, o: self._arith(o, lambda a, b: a - b)
    __rsub__ = lambda self, o: self._arith(o, lambda a, b: b - a)

    def expand(self, *shape):
        return from_numpy(np.broadcast_to(self.cpu().numpy(), shape).copy()).to(self.device)

    def __getitem__(self, key):
        """rows of the first dimension: an index or a unit-stride slice (a view of the same memory); [:, None]: a view with one more axis"""
        if isinstance(key, tuple):
            assert key == (slice(None), None), key
            return self._view((self.shape[0], 1) + self.shape[1:])
        n = self.shape[0]
        if isinstance(key, int):
            start, stop, shape = key % n, key % n + 1, self.shape[1:]
        else:
            start, stop, step = key.indices(n)
            assert step == 1, key
            shape = (max(stop - start, 0),) + self.shape[1:]
        row = (int(np.prod(self.shape[1:])) if len(self.shape) > 1 else 1) * self.element_size()
        if self._host is not None:
            return Tensor(shape, self.dtype, host=self._host[key] if not isinstance(key, int) else self._host[key:key + 1].reshape(shape), owner=self)
        return Tensor(shape, self.dtype, ptr=self._ptr + start * row, owner=self)

    def copy_(self, src, non_blocking=False):
        a = np.ascontiguousarray(src.cpu().numpy().astype(self.dtype.np, copy=False)).reshape(self.shape)
        if self.is_cuda:
            if a.nbytes:
                _check(_hip.hipMemcpy(self._ptr, a.ctypes.data, a.nbytes, 1), "hipMemcpy H2D")
        else:
            self._host[...] = a
        return self

    def pin_memory(self):
        return self


class _Block:
    def __init__(self, ptr):
        self.ptr = ptr

    def __del__(self):
        try:
            _hip.hipFree(self.ptr)
        except Exception:       # noqa: BLE001
            pass


def from_numpy(a):
    a = np.ascontiguousarray(a)
    return Tensor(a.shape, _BY_NP[a.dtype], host=a)


def as_tensor(x, dtype=None, device=None):
    t = x if isinstance(x, Tensor) else from_numpy(np.asarray(x, dtype=None if dtype is None else dtype.np))
    return t.to(device, dtype=dtype) if (device is not None or dtype is not None) else t


tensor = as_tensor


def empty(*shape, dtype=float32, device=None):
    shape = list(shape[0]) if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else list(shape)
    # (filled with a pattern, not zeros: a kernel that forgets to write something shows)
    host = np.full(shape, 0, dtype.np) if dtype.np.kind != "f" else np.full(shape, np.nan, dtype.np)
    t = from_numpy(host)
    return t.to(device) if device is not None else t


def zeros(*shape, dtype=float32, device=None):
    shape = list(shape[0]) if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else list(shape)
    t = from_numpy(np.zeros(shape, dtype.np))
    return t.to(device) if device is not None else t


def zeros_like(t):
    return zeros(*t.shape, dtype=t.dtype, device=t.device)


def empty_like(t):
    return empty(*t.shape, dtype=t.dtype, device=t.device)


def cumsum(x, dim, dtype=None, out=None):
    r = np.cumsum(x.cpu().numpy(), axis=dim, dtype=None if dtype is None else dtype.np)
    return out.copy_(from_numpy(r)) if out is not None else from_numpy(r).to(x.device)


def maximum(a, b, out=None):
    r = np.maximum(a.cpu().numpy(), b.cpu().numpy())
    return out.copy_(from_numpy(r)) if out is not None else from_numpy(r).to(a.device)


class OutOfMemoryError(RuntimeError):
    pass


def is_tensor(x):
    return isinstance(x, Tensor)


def is_autocast_enabled():
    return False


class _Stream:
    cuda_stream = 0

    def synchronize(self):
        _check(_hip.hipDeviceSynchronize(), "hipDeviceSynchronize")


class cuda:
    @staticmethod
    def current_stream(dev=None):
        return _Stream()

    @staticmethod
    def synchronize(dev=None):
        _check(_hip.hipDeviceSynchronize(), "hipDeviceSynchronize")

    @staticmethod
    def is_available():
        return True

    class Event:
        """the stand-in synchronises the device wherever an event is waited for or queried"""

        def __init__(self, enable_timing=False):
            pass

        def record(self, stream=None):
            pass

        def query(self):
            _check(_hip.hipDeviceSynchronize(), "hipDeviceSynchronize")
            return True

        def synchronize(self):
            _check(_hip.hipDeviceSynchronize(), "hipDeviceSynchronize")


device = _Device
dtype = _DType
bool = _bool
