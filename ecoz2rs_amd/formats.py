"""numpy readers/writers for the C-format files of the path.

Layouts: 16-byte ident + 96-byte class name (/root/reference/src/utl/mod.rs:19-41), then
  .prd    u32 T, u32 P, T*(P+1) f64          
  .cbook  u32 P, u32 M, M*(P+1) f64          (reflection coefficients)
  .seq    u32 T, u32 M, T*u16                (/root/reference/src/sequence/mod.rs:49-75)
"""
import os
import struct

import numpy as np

FILE_IDENT_LEN = 16
MAX_CLASS_NAME_LEN = 96


def _fixed(s, n):
    b = s.encode()[: n - 1]
    return b + b"\0" * (n - len(b))


def _read_fixed(b):
    # src/utl/mod.rs:30-41: up to the first NUL (or fixed_len - 1)
    pos = b.find(b"\0")
    if pos < 0:
        pos = len(b) - 1
    return b[:pos].decode()


def _header(f, ident, what):
    got = _read_fixed(f.read(FILE_IDENT_LEN))
    if not got.startswith(ident):
        raise ValueError(f"Not a {what}")
    return _read_fixed(f.read(MAX_CLASS_NAME_LEN))


def _write(path, ident, class_name, a, b, payload):
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    with open(path, "wb") as f:
        f.write(_fixed(ident, FILE_IDENT_LEN))
        f.write(_fixed(class_name, MAX_CLASS_NAME_LEN))
        f.write(struct.pack("<II", a, b))
        f.write(payload)


def write_prd(path, class_name, frames):
    frames = np.ascontiguousarray(frames, dtype="<f8")
    T, nc = frames.shape
    _write(path, "<predictor>", class_name, T, nc - 1, frames.tobytes())


def read_prd(path):
    with open(path, "rb") as f:
        cls = _header(f, "<predictor>", "predictor")
        T, P = struct.unpack("<II", f.read(8))
        data = np.frombuffer(f.read(T * (P + 1) * 8), dtype="<f8").reshape(T, P + 1)
    return cls, P, data.copy()


def write_cbook(path, class_name, reflections):
    reflections = np.ascontiguousarray(reflections, dtype="<f8")
    M, nc = reflections.shape
    _write(path, "<codebook>", class_name, nc - 1, M, reflections.tobytes())


def read_cbook(path):
    with open(path, "rb") as f:
        cls = _header(f, "<codebook>", "codebook")
        P, M = struct.unpack("<II", f.read(8))
        data = np.frombuffer(f.read(M * (P + 1) * 8), dtype="<f8").reshape(M, P + 1)
    return cls, P, data.copy()


def write_seq(path, class_name, M, symbols):
    symbols = np.ascontiguousarray(symbols, dtype="<u2")
    _write(path, "<sequence>", class_name, symbols.shape[0], M, symbols.tobytes())


def read_seq(path):
    """Follows Sequence::load, src/sequence/mod.rs:49-75, field by field."""
    with open(path, "rb") as f:
        cls = _header(f, "<sequence>", "sequence")
        T, M = struct.unpack("<II", f.read(8))
        sym = np.frombuffer(f.read(T * 2), dtype="<u2")
    return cls, M, sym.copy()
