"""Minimal reader / writer of Torch7's binary serialisation, for the one file of that format on the sampling path:
`<data>/test/val_captions.t7` = {raw_txt = {caption strings}, fea_txt = {FloatTensor (n_i, 1024) ...}} (the reference reads it
with the `torchfile` package, code/coco/stackgan/trainer.py:300-302; not installed here).

Format (little endian): an object is an int32 type tag followed by its payload --
  0 nil | 1 number (float64) | 2 string (int32 length, bytes) | 5 boolean (int32)
  3 table: int32 memo index; first occurrence: int32 count, then count (key object, value object) pairs
  4 torch object: int32 memo index; first occurrence: version string ("V 1", written like a string payload), class name
    (string payload), then for torch.<T>Tensor: int32 ndim, int64 sizes[ndim], int64 strides[ndim], int64 storage offset (1-based),
    a torch.<T>Storage object; for torch.<T>Storage: int64 count, raw elements.
Tables whose keys are 1..n come back as lists, other tables as `Table` (dict with attribute access), tensors as numpy arrays
(strides honoured), strings as `str` (utf-8, undecodable bytes replaced)."""
import struct

import numpy as np

_DT = {"Float": np.float32, "Double": np.float64, "Long": np.int64, "Int": np.int32, "Short": np.int16, "Byte": np.uint8,
       "Char": np.int8}


class Table(dict):
    __getattr__ = dict.__getitem__


class _Reader(object):
    def __init__(self, f):
        self.f, self.memo = f, {}

    def _raw(self, fmt):
        n = struct.calcsize(fmt)
        b = self.f.read(n)
        if len(b) != n:
            raise EOFError("truncated t7 file")
        return struct.unpack("<" + fmt, b)

    def _bytes(self, n):
        if n < 0:
            raise ValueError("t7: negative length %d" % n)
        b = self.f.read(n)
        if len(b) != n:
            raise EOFError("truncated t7 file")
        return b

    def _string(self):
        (n,) = self._raw("i")
        return self._bytes(n)

    def obj(self):
        (tag,) = self._raw("i")
        if tag == 0:
            return None
        if tag == 1:
            v = self._raw("d")[0]
            return int(v) if float(v).is_integer() else v
        if tag == 2:
            return self._string().decode("utf-8", "replace")
        if tag == 5:
            return self._raw("i")[0] == 1
        if tag == 3:
            (idx,) = self._raw("i")
            if idx in self.memo:
                return self.memo[idx]
            (n,) = self._raw("i")
            t = Table()
            self.memo[idx] = t            # (a table that refers to ITSELF keeps this Table object inside; the caller gets the
            #                               list form below -- cycles do not occur in the files this reader is for, val_captions.t7)
            for _ in range(n):
                k = self.obj()
                t[k] = self.obj()
            if n and all(isinstance(k, int) for k in t) and sorted(t) == list(range(1, n + 1)):
                lst = [t[i] for i in range(1, n + 1)]
                self.memo[idx] = lst
                return lst
            return t
        if tag == 4:
            (idx,) = self._raw("i")
            if idx in self.memo:
                return self.memo[idx]
            ver = self._string()
            cls = self._string().decode() if ver.startswith(b"V ") else ver.decode()
            kind = cls.split(".")[-1]
            if kind.endswith("Storage"):
                (n,) = self._raw("q")
                dt = np.dtype(_DT[kind[:-7]])
                a = np.frombuffer(self._bytes(n * dt.itemsize), dtype=dt).copy()
            elif kind.endswith("Tensor"):
                (nd,) = self._raw("i")
                size = self._raw("%dq" % nd) if nd else ()
                stride = self._raw("%dq" % nd) if nd else ()
                (off,) = self._raw("q")
                st = self.obj()
                if st is None or nd == 0:
                    a = np.zeros(size, dtype=_DT[kind[:-6]])
                else:
                    a = np.lib.stride_tricks.as_strided(st[off - 1:], shape=size,
                                                        strides=[s * st.itemsize for s in stride]).copy()
            else:
                raise ValueError("t7: unsupported torch class %s" % cls)
            self.memo[idx] = a
            return a
        raise ValueError("t7: unsupported type tag %d" % tag)


def load(path):
    with open(path, "rb") as f:
        return _Reader(f).obj()


class _Writer(object):
    def __init__(self, f):
        self.f, self.next = f, 1

    def _string(self, s):
        b = s if isinstance(s, bytes) else s.encode("utf-8")
        self.f.write(struct.pack("<i", len(b)) + b)

    def obj(self, o):
        w = self.f.write
        if o is None:
            w(struct.pack("<i", 0))
        elif isinstance(o, bool):
            w(struct.pack("<ii", 5, int(o)))
        elif isinstance(o, (int, float)):
            w(struct.pack("<id", 1, float(o)))
        elif isinstance(o, (str, bytes)):
            w(struct.pack("<i", 2))
            self._string(o)
        elif isinstance(o, (list, tuple, dict)):
            items = list(o.items()) if isinstance(o, dict) else [(i + 1, v) for i, v in enumerate(o)]
            w(struct.pack("<iii", 3, self.next, len(items)))
            self.next += 1
            for k, v in items:
                self.obj(k)
                self.obj(v)
        elif isinstance(o, np.ndarray):
            name = {np.dtype(v): k for k, v in _DT.items()}[o.dtype]
            a = np.ascontiguousarray(o)
            w(struct.pack("<ii", 4, self.next))
            self.next += 1
            self._string("V 1")
            self._string("torch.%sTensor" % name)
            strides = [int(s // a.itemsize) for s in a.strides]
            w(struct.pack("<i", a.ndim) + struct.pack("<%dq" % a.ndim, *a.shape) + struct.pack("<%dq" % a.ndim, *strides) +
              struct.pack("<q", 1))
            w(struct.pack("<ii", 4, self.next))
            self.next += 1
            self._string("V 1")
            self._string("torch.%sStorage" % name)
            w(struct.pack("<q", a.size) + a.tobytes())
        else:
            raise TypeError("t7: cannot write %r" % type(o))


def save(path, obj):
    """the inverse of `load` for nil / numbers / strings / booleans / lists / dicts / numpy arrays (what a caption file holds)."""
    with open(path, "wb") as f:
        _Writer(f).obj(obj)
