"""Minimal NRRD (Nearly Raw Raster Data, teem.sourceforge.net/nrrd/format.html) reader / writer — what
`nrrd.read(path)` gives the reference's volume reader (dataset/few_shot_reader.py:314,321: `data, header = nrrd.read(f)`).

The reference depends on the third-party `pynrrd` package (not vendored, not installed here).  This module follows
the published file format, not pynrrd's source: magic line `NRRD000x`, `field: value` header lines up to the first
blank line, then the payload (`encoding`: raw, gzip / gz, bzip2 / bz2, ascii / text / txt; `endian`; `byte skip`,
`line skip`; attached payload or a single detached `data file`).  Like pynrrd's default (index_order='F') the array
comes back with `shape == sizes`, i.e. the FIRST header axis is the fastest one in the file, and the header as a dict
(`sizes` an int array, `type`, `dimension`, `encoding`, `endian`, every other field as its string; `key:=value`
pairs as strings).  Parity with pynrrd itself is unpinned (the package is absent); the tests hold the reader to files
written by an independent path (hand-written headers + numpy / gzip / bz2 payloads, both byte orders, attached and
detached), the writer to an independent parser, and both to round trips.
"""
import bz2
import gzip
import os
import zlib

import numpy as np

_TYPES = {}
for _names, _code in [(("signed char", "int8", "int8_t"), "i1"), (("uchar", "unsigned char", "uint8", "uint8_t"), "u1"),
                      (("short", "short int", "signed short", "signed short int", "int16", "int16_t"), "i2"),
                      (("ushort", "unsigned short", "unsigned short int", "uint16", "uint16_t"), "u2"),
                      (("int", "signed int", "int32", "int32_t"), "i4"), (("uint", "unsigned int", "uint32", "uint32_t"), "u4"),
                      (("longlong", "long long", "long long int", "signed long long", "signed long long int", "int64",
                        "int64_t"), "i8"),
                      (("ulonglong", "unsigned long long", "unsigned long long int", "uint64", "uint64_t"), "u8"),
                      (("float",), "f4"), (("double",), "f8")]:
    for _n in _names:
        _TYPES[_n] = _code
_NAMES = {"i1": "int8", "u1": "uint8", "i2": "int16", "u2": "uint16", "i4": "int32", "u4": "uint32", "i8": "int64",
          "u8": "uint64", "f4": "float", "f8": "double"}


class NRRDError(ValueError):
    pass


def _parse_header(f):
    magic = f.readline().decode("ascii", "replace").rstrip("\r\n")
    if not (magic.startswith("NRRD000") and magic[7:].isdigit() and 1 <= int(magic[7:]) <= 5):
        raise NRRDError(f"not an NRRD file (magic line {magic!r})")
    header = {}
    while True:
        raw = f.readline()
        if not raw:
            break                                   # header only (detached payload)
        line = raw.decode("ascii", "replace").rstrip("\r\n")
        if line == "":
            break
        if line.startswith("#"):
            continue
        if ":=" in line and (": " not in line or line.index(":=") < line.index(": ")):
            k, v = line.split(":=", 1)
            header[k] = v
            continue
        if ": " not in line and not line.endswith(":"):
            raise NRRDError(f"malformed header line {line!r}")
        k, _, v = line.partition(":")
        header[k.strip().lower()] = v.strip()
    for alias, name in [("datafile", "data file"), ("lineskip", "line skip"), ("byteskip", "byte skip")]:
        if alias in header:
            header[name] = header.pop(alias)
    for need in ("type", "dimension", "sizes", "encoding"):
        if need not in header:
            raise NRRDError(f"header lacks the required field '{need}'")
    header["dimension"] = int(header["dimension"])
    header["sizes"] = np.array([int(s) for s in header["sizes"].split()], dtype=int)
    if len(header["sizes"]) != header["dimension"] or (header["sizes"] <= 0).any():
        raise NRRDError(f"sizes {header['sizes'].tolist()} do not fit dimension {header['dimension']}")
    return header


def _dtype(header):
    t = header["type"].strip().lower()
    if t == "block":
        raise NRRDError("type 'block' is not supported")
    if t not in _TYPES:
        raise NRRDError(f"unknown type {header['type']!r}")
    code = _TYPES[t]
    if code[1] == "1":
        return np.dtype(code)
    endian = header.get("endian", "").lower()
    if endian not in ("little", "big"):
        raise NRRDError("multi-byte type without a valid 'endian' field")
    return np.dtype(("<" if endian == "little" else ">") + code)


def read(filename):
    """-> (data, header); data.shape == header['sizes'] (first axis fastest in the file), native byte order."""
    with open(filename, "rb") as f:
        header = _parse_header(f)
        dt = _dtype(header)
        if "data file" in header:
            name = header["data file"]
            if name.upper().startswith("LIST") or "%" in name:
                raise NRRDError("multi-file detached payloads are not supported")
            path = name if os.path.isabs(name) else os.path.join(os.path.dirname(os.path.abspath(filename)), name)
            with open(path, "rb") as g:
                payload = g.read()
        else:
            payload = f.read()
    n = int(np.prod(header["sizes"]))
    enc = header["encoding"].lower()
    line_skip, byte_skip = int(header.get("line skip", 0)), int(header.get("byte skip", 0))
    if enc in ("gzip", "gz"):
        # line skip / byte skip >= 0 apply to the DEcompressed stream for compressed encodings
        payload = zlib.decompress(payload, zlib.MAX_WBITS | 16) if payload[:2] == b"\x1f\x8b" else zlib.decompress(payload)
    elif enc in ("bzip2", "bz2"):
        payload = bz2.decompress(payload)
    elif enc not in ("raw", "ascii", "text", "txt"):
        raise NRRDError(f"unsupported encoding {header['encoding']!r}")
    for _ in range(line_skip):
        payload = payload[payload.index(b"\n") + 1:]
    if enc in ("ascii", "text", "txt"):
        flat = np.array(payload[max(byte_skip, 0):].split(), dtype=np.float64 if dt.kind == "f" else np.int64).astype(dt.newbyteorder("="))
    else:
        if byte_skip == -1:
            payload = payload[len(payload) - n * dt.itemsize:]
        elif byte_skip > 0:
            payload = payload[byte_skip:]
        if len(payload) < n * dt.itemsize:
            raise NRRDError(f"payload holds {len(payload)} bytes, sizes need {n * dt.itemsize}")
        flat = np.frombuffer(payload, dtype=dt, count=n).astype(dt.newbyteorder("="))
    if flat.size != n:
        raise NRRDError(f"payload holds {flat.size} values, sizes need {n}")
    return flat.reshape(tuple(int(s) for s in header["sizes"]), order="F"), header


def write(filename, data, header=None, encoding="gzip"):
    """data [sizes...] -> NRRD0004 file with an attached payload (first axis fastest), little endian."""
    data = np.asarray(data)
    code = data.dtype.str[1:]
    if code not in _NAMES:
        raise NRRDError(f"dtype {data.dtype} has no NRRD type")
    fields = {"type": _NAMES[code], "dimension": str(data.ndim), "sizes": " ".join(str(s) for s in data.shape)}
    if code[1] != "1":
        fields["endian"] = "little"
    fields["encoding"] = encoding
    extra = {k: v for k, v in (header or {}).items() if k not in fields and k not in ("data file", "line skip", "byte skip")}
    payload = np.asfortranarray(data.astype(data.dtype.newbyteorder("<"))).tobytes(order="F")
    if encoding in ("gzip", "gz"):
        payload = gzip.compress(payload, compresslevel=4, mtime=0)
    elif encoding in ("bzip2", "bz2"):
        payload = bz2.compress(payload)
    elif encoding != "raw":
        raise NRRDError(f"unsupported encoding {encoding!r}")
    with open(filename, "wb") as f:
        f.write(b"NRRD0004\n# written by rpnet_amd.utils.nrrd\n")
        for k, v in list(fields.items()) + list(extra.items()):
            f.write(f"{k}: {v}\n".encode("ascii"))
        f.write(b"\n")
        f.write(payload)
