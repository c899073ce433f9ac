"""CPU ORACLE (test infrastructure, NOT product code): pure-Python restatement of the reference's
PCD reader / writer, pc/io.go:33-285 (Unmarshal, unmarshalPCDHeaderTo, unmarshalPCDDataTo, Marshal).

Pinned by the reference's own fixtures (pc/io_test.go, transcribed in tests/golden/ref_pcd.json) via
tests/test_oracle_golden.py.  The LZF decoder restates the published liblzf format that the
reference's only third-party module implements -- github.com/zhuyie/golzf
v0.0.0-20161112031142-8387b0307ade (go.mod:5), absent from /root/reference -- anchored on the
reference's BinaryCompressed fixture and its ErrDataCorruption case.

Quirk kept (pc/io.go:208-227): the binary_compressed de-interleave copies Size[i] bytes per field
from head[i] + p*Size[i] -- COUNT is ignored on both sides, so only element 0 of a COUNT > 1
field is filled (from the first third of the field's block) and the rest stays zero."""
import struct

import numpy as np


class PcdError(ValueError):
    """kind: 'syntax' (strconv.ErrSyntax), 'eof' (io.EOF / ErrUnexpectedEOF), 'corrupt'
    (lzf.ErrDataCorruption), 'header' (the errors.New cases of io.go:55,119,125-133,202)."""

    def __init__(self, kind, msg):
        self.kind = kind
        super().__init__(msg)


def lzf_decompress(src, out_len):
    """liblzf lzf_decompress: control byte < 32: literal run of ctrl+1 bytes; else a back reference
    of length (ctrl >> 5) + 2 (+ next byte if the 3-bit length is 7) at distance
    ((ctrl & 0x1f) << 8 | next byte) + 1.  Returns the number of bytes written."""
    out = bytearray(out_len)
    ip, op, n = 0, 0, len(src)
    while ip < n:
        ctrl = src[ip]
        ip += 1
        if ctrl < 32:
            ctrl += 1
            if op + ctrl > out_len:
                raise PcdError("corrupt", "lzf: output too small")
            if ip + ctrl > n:
                raise PcdError("corrupt", "lzf: data corruption")
            out[op:op + ctrl] = src[ip:ip + ctrl]
            ip += ctrl
            op += ctrl
        else:
            ln = ctrl >> 5
            ref = op - ((ctrl & 0x1F) << 8) - 1
            if ip >= n:
                raise PcdError("corrupt", "lzf: data corruption")
            if ln == 7:
                ln += src[ip]
                ip += 1
                if ip >= n:
                    raise PcdError("corrupt", "lzf: data corruption")
            ref -= src[ip]
            ip += 1
            ln += 2
            if op + ln > out_len:
                raise PcdError("corrupt", "lzf: output too small")
            if ref < 0:
                raise PcdError("corrupt", "lzf: data corruption")
            for _ in range(ln):  # byte by byte: the ranges may overlap
                out[op] = out[ref]
                op += 1
                ref += 1
    return bytes(out), op


def _atoi(s):
    try:
        if not s or s.strip() != s or not (s.lstrip("+-").isdigit()):
            raise ValueError
        return int(s)
    except ValueError:
        raise PcdError("syntax", "strconv.Atoi: parsing %r: invalid syntax" % s)


def _parse_float32(s):
    try:
        if "_" in s or s.strip() != s:
            raise ValueError
        return np.float32(float(s))
    except ValueError:
        raise PcdError("syntax", "strconv.ParseFloat: parsing %r: invalid syntax" % s)


def _read_line(buf, pos):
    """bufio.Reader.ReadLine: up to '\\n' (a trailing '\\r' dropped); None at EOF."""
    if pos >= len(buf):
        return None, pos
    e = buf.find(b"\n", pos)
    if e < 0:
        return buf[pos:], len(buf)
    line = buf[pos:e]
    if line.endswith(b"\r"):
        line = line[:-1]
    return line, e + 1


def unmarshal_header(buf):
    """io.go:47-136 -> (header dict, nPoints, format, offset of the data)."""
    h = dict(version=np.float32(0), fields=[], size=[], type=[], count=[], width=0, height=0, viewpoint=[])
    npoints, fmt, pos = 0, None, 0
    while True:
        line, pos = _read_line(buf, pos)
        if line is None:
            raise PcdError("eof", "EOF")
        args = line.decode("latin-1").split()
        if len(args) < 2:
            raise PcdError("header", "header field must have value")
        k = args[0]
        if k == "VERSION":
            h["version"] = _parse_float32(args[1])
        elif k == "FIELDS":
            h["fields"] = args[1:]
        elif k == "SIZE":
            h["size"] = [_atoi(a) for a in args[1:]]
        elif k == "TYPE":
            h["type"] = args[1:]
        elif k == "COUNT":
            h["count"] = [_atoi(a) for a in args[1:]]
        elif k == "WIDTH":
            h["width"] = _atoi(args[1])
        elif k == "HEIGHT":
            h["height"] = _atoi(args[1])
        elif k == "VIEWPOINT":
            h["viewpoint"] = [_parse_float32(a) for a in args[1:]]
        elif k == "POINTS":
            npoints = _atoi(args[1])
        elif k == "DATA":
            if args[1] not in ("ascii", "binary", "binary_compressed"):
                raise PcdError("header", "unknown data format")
            fmt = args[1]
            break
    if len(h["fields"]) != len(h["size"]):
        raise PcdError("header", "size field size is wrong")
    if len(h["fields"]) != len(h["type"]):
        raise PcdError("header", "type field size is wrong")
    if len(h["fields"]) != len(h["count"]):
        raise PcdError("header", "count field size is wrong")
    return h, npoints, fmt, pos


def stride_of(h):
    return sum(s * c for s, c in zip(h["size"], h["count"]))  # pointcloud.go:64-70


def unmarshal(buf):
    """io.go:33-45,138-230 -> (header dict, points, data bytes)."""
    buf = bytes(buf)
    h, n, fmt, pos = unmarshal_header(buf)
    stride = stride_of(h)
    if fmt == "ascii":
        data = bytearray(n * stride)
        off = 0
        while True:
            line, pos = _read_line(buf, pos)
            if line is None:
                break
            toks = line.decode("latin-1").split()
            lo = 0
            for i, ty in enumerate(h["type"]):
                for j in range(h["count"][i]):
                    if ty == "F":
                        data[off:off + 4] = struct.pack("<f", _parse_float32(toks[lo + j]))
                    elif ty == "U":
                        t = toks[lo + j]
                        if not t.isdigit() or int(t) >= 1 << 32:
                            raise PcdError("syntax", "strconv.ParseUint: parsing %r: invalid syntax" % t)
                        data[off:off + 4] = struct.pack("<I", int(t))
                    off += h["size"][i]
                lo += h["count"][i]
        return h, n, bytes(data)
    if fmt == "binary":
        need = n * stride
        if len(buf) - pos < need:
            raise PcdError("eof", "EOF")
        return h, n, buf[pos:pos + need]
    if len(buf) - pos < 4:
        raise PcdError("eof", "EOF")
    (ncomp,) = struct.unpack_from("<i", buf, pos)
    pos += 4
    if len(buf) - pos < 4:
        raise PcdError("eof", "EOF")
    (nunc,) = struct.unpack_from("<i", buf, pos)
    pos += 4
    if len(buf) - pos < ncomp:
        raise PcdError("eof", "EOF")
    dec, got = lzf_decompress(buf[pos:pos + ncomp], nunc)
    if got != nunc:
        raise PcdError("header", "wrong uncompressed size")
    head, offset, p, o = [], [], 0, 0
    for s, c in zip(h["size"], h["count"]):
        head.append(p)
        offset.append(o)
        p += s * c * n
        o += s * c
    data = bytearray(got)
    for pt in range(n):
        for i in range(len(head)):
            size = h["size"][i]
            to = pt * stride + offset[i]
            frm = head[i] + pt * size  # sic: COUNT ignored (io.go:222)
            data[to:to + size] = dec[frm:frm + size]
    return h, n, bytes(data)


def marshal(h, points, data):
    """io.go:232-285: always DATA binary; missing viewpoint -> 0 0 0 1 0 0 0."""
    vp = list(h["viewpoint"]) or [0, 0, 0, 1, 0, 0, 0]
    head = "VERSION %0.1f\nFIELDS %s\nSIZE %s\nTYPE %s\nCOUNT %s\nWIDTH %d\nHEIGHT %d\nVIEWPOINT %s\nPOINTS %d\nDATA binary\n" % (
        float(h["version"]), " ".join(h["fields"]), " ".join(str(s) for s in h["size"]), " ".join(h["type"]),
        " ".join(str(c) for c in h["count"]), h["width"], h["height"],
        " ".join("%.4f" % float(np.float32(v)) for v in vp), points)
    return head.encode() + bytes(data)
