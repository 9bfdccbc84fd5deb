"""Test-harness allocator: host arrays that END at an inaccessible page.

Native code that writes one byte past an out-array handed over by the bindings faults at that write (SIGSEGV inside
the offending call, with a native frame on the stack) instead of silently damaging whatever the allocator placed
behind the array.  conftest.py installs `empty` as the bindings' `alloc_out`."""
import ctypes
import mmap

import numpy as np

_PAGE = mmap.PAGESIZE
_libc = ctypes.CDLL(None, use_errno=True)
_libc.mprotect.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
_libc.mprotect.restype = ctypes.c_int


def empty(shape, dtype):
    dtype = np.dtype(dtype)
    count = int(np.prod(shape)) if not np.isscalar(shape) else int(shape)
    nbytes = count * dtype.itemsize
    body = max(_PAGE, -(-nbytes // _PAGE) * _PAGE)
    mm = mmap.mmap(-1, body + _PAGE)                       # zero-filled, private, anonymous
    base = ctypes.addressof(ctypes.c_char.from_buffer(mm))
    if _libc.mprotect(base + body, _PAGE, 0) != 0:          # PROT_NONE on the page behind the array
        raise OSError(ctypes.get_errno(), "mprotect")
    # start so that the last element ends exactly at the guard page; keep the dtype's alignment
    start = body - nbytes
    start -= start % max(dtype.alignment, 4)
    arr = np.frombuffer(mm, dtype=dtype, count=count, offset=start)
    return arr.reshape(shape)
