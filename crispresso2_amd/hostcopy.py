"""Device tensor <-> numpy array through pinned staging memory this module keeps.

Why not tensor.cpu(): a device-to-host copy into PAGEABLE memory makes the HIP runtime lock the destination pages for the transfer;
measured on the MI355X box (profiles/r03/README.md, "Ingest"): after three 28 MB copies of that kind the next small copy -- an .item() --
took 20-35 ms, every time.  Through a pinned buffer the same three arrays cost their transfer (0.5 ms each at the link's rate) plus one
host memcpy each, and nothing later pays for them."""
import threading

import numpy as np

_staging = {}
_lock = threading.Lock()                # one staging buffer per device: a copy at a time


def to_host(t, dtype=None):
    """-> a numpy array that owns its memory (dtype: converted on the host while copying out of the staging buffer).  Waits for the
    current stream of the tensor's device."""
    import torch
    if t.device.type != "cuda":
        a = t.numpy()
        return a.astype(dtype) if dtype is not None else a.copy()
    t = t.contiguous()
    nbytes = t.numel() * t.element_size()
    if nbytes == 0:
        return np.zeros(tuple(t.shape), dtype=dtype or t.numpy(force=True).dtype)
    key = t.device.index
    with _lock:
        buf = _staging.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(max(nbytes, 64 << 20), dtype=torch.uint8, pin_memory=True)
            _staging[key] = buf
        view = buf[:nbytes].view(t.dtype).view(t.shape)
        view.copy_(t, non_blocking=True)
        torch.cuda.current_stream(t.device).synchronize()
        a = view.numpy()
        return a.astype(dtype) if dtype is not None else a.copy()


def release_staging():
    """give the pinned staging buffers back (at least 64 MiB per device, grown to the largest array that went through)"""
    with _lock:
        _staging.clear()


def to_device(a, dev):
    """numpy array -> tensor on `dev` (same dtype and shape), staged through the pinned buffer: the array's own (pageable, freshly mapped)
    pages are never registered with the driver.  Small arrays and non-GPU devices: torch's own copy."""
    import torch
    a = np.ascontiguousarray(a)
    if dev.type != "cuda" or a.nbytes < (1 << 20):
        return torch.from_numpy(a if a.flags.writeable else a.copy()).to(dev)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    with _lock:
        buf = _staging.get(key)
        if buf is None or buf.numel() < a.nbytes:
            buf = torch.empty(max(a.nbytes, 64 << 20), dtype=torch.uint8, pin_memory=True)
            _staging[key] = buf
        flat = a.reshape(-1).view(np.uint8)
        buf.numpy()[:a.nbytes] = flat
        out = torch.empty(a.nbytes, dtype=torch.uint8, device=dev)
        out.copy_(buf[:a.nbytes], non_blocking=True)
        torch.cuda.current_stream(dev).synchronize()                  # (the staging buffer is free again)
    return out.view(torch.from_numpy(np.zeros(1, dtype=a.dtype)).dtype).view(a.shape)
