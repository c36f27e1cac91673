"""OSCD ingest: the reference's utils/dataloaders.py:51-145 (get_train_val_metadata, label_loader, city_loader,
full_onera_loader) without rasterio / cv2.

* File decoding is host work and stays on the host: a small baseline-TIFF reader (strips or tiles; no compression,
  PackBits, LZW, Deflate; horizontal predictor; 8/16/32-bit single-sample rasters -- what Sentinel-2 band files are)
  and a PNG reader for the change masks (8-bit gray / RGB / RGBA / palette, non-interlaced; colour is reduced with
  cv2's fixed-point luma like ``cv2.imread(path, 0)``).
* The arithmetic -- per-band ``(band - mean) / std`` and ``cv2.resize`` to the label grid -- runs on the device
  (``bdn_ingest_band``), one launch per band, straight into the band's plane of the ``[2, C, H, W]`` city stack, which is
  the layout ``predict_scene`` / ``bdn_gather_tiles`` consume.  A 10 980 x 10 980 Sentinel-2 tile is 13 launches per date.

PARITY: rasterio / cv2 are not installed in the build image, so this module is pinned against the written-out
algorithm (oracle/ingest_oracle.py) and against files produced by the writers below, not against the reference's own
output.  One stated deviation: the reference builds ``train_cities`` from a ``set`` difference (utils/dataloaders.py:55),
whose order is not deterministic; here the training cities keep their sorted order.
"""
import glob
import os
import struct
import zlib

import numpy as np
import torch

from .. import _lib
from .._lib import call, ptr
from .dataloaders import patch_origins

# --------------------------------------------------------------------------------------------------------------- TIFF
_TYPES = {1: 'B', 2: 'c', 3: 'H', 4: 'I', 5: 'II', 6: 'b', 8: 'h', 9: 'i', 11: 'f', 12: 'd', 16: 'Q'}


def _lzw_decode(data):
    """TIFF LZW (MSB-first codes, 9..12 bits, ClearCode 256, EOI 257, early change)."""
    out = bytearray()
    table = [bytes([i]) for i in range(256)] + [b'', b'']
    nbits, bitbuf, bitcnt, prev = 9, 0, 0, None
    for byte in data:
        bitbuf = (bitbuf << 8) | byte
        bitcnt += 8
        while bitcnt >= nbits:
            code = (bitbuf >> (bitcnt - nbits)) & ((1 << nbits) - 1)
            bitcnt -= nbits
            if code == 256:
                table = table[:258]
                nbits, prev = 9, None
                continue
            if code == 257:
                return bytes(out)
            if prev is None:
                entry = table[code]
            else:
                entry = table[code] if code < len(table) else prev + prev[:1]
                table.append(prev + entry[:1])
            out += entry
            prev = entry
            if len(table) + 1 >= (1 << nbits) and nbits < 12:
                nbits += 1
    return bytes(out)


def _packbits_decode(data):
    out, i = bytearray(), 0
    while i < len(data):
        n = data[i]
        i += 1
        if n < 128:
            out += data[i:i + n + 1]
            i += n + 1
        elif n > 128:
            out += data[i:i + 1] * (257 - n)
            i += 1
    return bytes(out)


def read_tiff(path):
    """First image of a baseline TIFF as a 2-D numpy array (what ``rasterio.open(path).read()[0]`` returns for the
    single-band Sentinel-2 files of OSCD, utils/dataloaders.py:92)."""
    with open(path, 'rb') as fh:
        buf = fh.read()
    bo = {b'II': '<', b'MM': '>'}.get(buf[:2])
    if bo is None or struct.unpack(bo + 'H', buf[2:4])[0] != 42:
        raise ValueError(f'{path}: not a classic TIFF (BigTIFF is not supported)')
    off = struct.unpack(bo + 'I', buf[4:8])[0]
    tags = {}
    for k in range(struct.unpack(bo + 'H', buf[off:off + 2])[0]):
        tag, typ, cnt, val = struct.unpack(bo + 'HHI4s', buf[off + 2 + 12 * k: off + 14 + 12 * k])
        fmt = _TYPES.get(typ)
        if fmt is None:
            continue
        size = struct.calcsize(bo + fmt) * cnt
        raw = val[:size] if size <= 4 else buf[struct.unpack(bo + 'I', val)[0]:][:size]
        tags[tag] = struct.unpack(bo + fmt * cnt, raw) if typ != 2 else raw
    w, h = tags[256][0], tags[257][0]
    bits = tags.get(258, (1,))[0]
    spp = tags.get(277, (1,))[0]
    comp = tags.get(259, (1,))[0]
    fmt_tag = tags.get(339, (1,))[0]
    pred = tags.get(317, (1,))[0]
    if spp != 1:
        raise ValueError(f'{path}: {spp} samples per pixel; OSCD band files hold one')
    dt = {(8, 1): 'u1', (16, 1): 'u2', (32, 1): 'u4', (8, 2): 'i1', (16, 2): 'i2', (32, 2): 'i4', (32, 3): 'f4'}.get((bits, fmt_tag))
    if dt is None:
        raise ValueError(f'{path}: unsupported sample format {bits} bits / format {fmt_tag}')
    dt = np.dtype(bo + dt)

    def decode(chunk, rows, cols):
        if comp == 5:
            chunk = _lzw_decode(chunk)
        elif comp in (8, 32946):
            chunk = zlib.decompress(chunk)
        elif comp == 32773:
            chunk = _packbits_decode(chunk)
        elif comp != 1:
            raise ValueError(f'{path}: TIFF compression {comp} is not supported')
        a = np.frombuffer(chunk, dtype=dt, count=rows * cols).reshape(rows, cols)
        if pred == 2:
            a = np.cumsum(a.astype(np.int64), axis=1).astype(dt.newbyteorder('='))
        return a

    img = np.empty((h, w), dtype=dt.newbyteorder('='))
    if 322 in tags:                                         # tiled
        tw, th = tags[322][0], tags[323][0]
        offs, cnts = tags[324], tags[325]
        per_row = (w + tw - 1) // tw
        for t, (o, c) in enumerate(zip(offs, cnts)):
            ty, tx = (t // per_row) * th, (t % per_row) * tw
            tile = decode(buf[o:o + c], th, tw)
            img[ty:ty + th, tx:tx + tw] = tile[:h - ty, :w - tx]
    else:
        rps = min(tags.get(278, (h,))[0], h)
        for s, (o, c) in enumerate(zip(tags[273], tags[279])):
            r0 = s * rps
            rows = min(rps, h - r0)
            img[r0:r0 + rows] = decode(buf[o:o + c], rows, w)
    return img


def write_tiff(path, arr, compression='none', rows_per_strip=64):
    """Minimal little-endian strip TIFF writer (tests and synthetic datasets): 'none' or 'deflate'."""
    arr = np.ascontiguousarray(arr)
    h, w = arr.shape
    fmt = {'u': 1, 'i': 2, 'f': 3}[arr.dtype.kind]
    strips = []
    for r0 in range(0, h, rows_per_strip):
        raw = arr[r0:r0 + rows_per_strip].astype(arr.dtype.newbyteorder('<')).tobytes()
        strips.append(zlib.compress(raw) if compression == 'deflate' else raw)
    n = len(strips)
    entries = [(256, 4, 1, w), (257, 4, 1, h), (258, 3, 1, arr.itemsize * 8), (259, 3, 1, 8 if compression == 'deflate' else 1),
               (262, 3, 1, 1), (273, 4, n, None), (277, 3, 1, 1), (278, 4, 1, rows_per_strip), (279, 4, n, None), (339, 3, 1, fmt)]
    ifd_off = 8
    ifd_size = 2 + 12 * len(entries) + 4
    arrays_off = ifd_off + ifd_size
    data_off = arrays_off + (8 * n if n > 1 else 0)
    offs, pos = [], data_off
    for s in strips:
        offs.append(pos)
        pos += len(s)
    out = bytearray(b'II' + struct.pack('<HI', 42, ifd_off))
    out += struct.pack('<H', len(entries))
    for tag, typ, cnt, val in entries:
        if val is None:
            vals = offs if tag == 273 else [len(s) for s in strips]
            val = vals[0] if n == 1 else arrays_off + (0 if tag == 273 else 4 * n)
        out += struct.pack('<HHII', tag, typ, cnt, val) if typ == 4 else struct.pack('<HHIHH', tag, typ, cnt, val, 0)
    out += struct.pack('<I', 0)
    if n > 1:
        out += struct.pack(f'<{n}I', *offs) + struct.pack(f'<{n}I', *[len(s) for s in strips])
    for s in strips:
        out += s
    with open(path, 'wb') as fh:
        fh.write(out)


# ---------------------------------------------------------------------------------------------------------------- PNG
def read_png_gray(path):
    """``cv2.imread(path, 0)``: 8-bit grayscale view of a non-interlaced PNG (gray, RGB, RGBA, palette)."""
    with open(path, 'rb') as fh:
        buf = fh.read()
    if buf[:8] != b'\x89PNG\r\n\x1a\n':
        raise ValueError(f'{path}: not a PNG')
    pos, idat, plte, hdr = 8, b'', None, None
    while pos < len(buf):
        n, kind = struct.unpack('>I4s', buf[pos:pos + 8])
        body = buf[pos + 8:pos + 8 + n]
        if kind == b'IHDR':
            hdr = struct.unpack('>IIBBBBB', body)
        elif kind == b'PLTE':
            plte = np.frombuffer(body, dtype=np.uint8).reshape(-1, 3)
        elif kind == b'IDAT':
            idat += body
        elif kind == b'IEND':
            break
        pos += 12 + n
    w, h, depth, ctype, _, _, interlace = hdr
    if depth != 8 or interlace:
        raise ValueError(f'{path}: only 8-bit non-interlaced PNGs are supported')
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    raw = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(h, 1 + w * ch)
    out = np.zeros((h, w * ch), dtype=np.uint8)
    prev = np.zeros(w * ch, dtype=np.int64)
    for y in range(h):
        f, line = int(raw[y, 0]), raw[y, 1:].astype(np.int64)
        if f == 2:
            line = line + prev
        elif f in (1, 3, 4):
            rec = np.zeros(w * ch + ch, dtype=np.int64)      # ch zeros of left context
            for x in range(w * ch):                          # sub / average / paeth need the running left neighbour
                a, b = rec[x], prev[x]
                c = prev[x - ch] if x >= ch else 0
                if f == 1:
                    p = a
                elif f == 3:
                    p = (a + b) >> 1
                else:
                    pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                    p = a if pa <= pb and pa <= pc else (b if pb <= pc else c)
                rec[x + ch] = (line[x] + p) & 255
            line = rec[ch:]
        line = line & 255
        out[y] = line
        prev = line
    px = out.reshape(h, w, ch)
    if ctype == 3:
        px = plte[px[..., 0]]
        ctype = 2
    if ctype in (0, 4):
        return px[..., 0].copy()
    r, g, b = (px[..., i].astype(np.int64) for i in range(3))   # cv2's fixed-point BT.601 luma (R2Y 4899, G2Y 9617, B2Y 1868)
    return ((r * 4899 + g * 9617 + b * 1868 + (1 << 13)) >> 14).astype(np.uint8)


def write_png_gray(path, arr):
    """8-bit grayscale PNG writer (tests and synthetic datasets)."""
    arr = np.ascontiguousarray(arr, dtype=np.uint8)
    h, w = arr.shape

    def chunk(kind, body):
        return struct.pack('>I', len(body)) + kind + body + struct.pack('>I', zlib.crc32(kind + body) & 0xffffffff)

    raw = b''.join(b'\x00' + arr[y].tobytes() for y in range(h))
    with open(path, 'wb') as fh:
        fh.write(b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, 8, 0, 0, 0, 0))
                 + chunk(b'IDAT', zlib.compress(raw)) + chunk(b'IEND', b''))


# ------------------------------------------------------------------------------------------- the reference's loader API
def _cities(data_dir):
    return sorted(i for i in os.listdir(data_dir + 'labels/')
                  if not i.startswith('.') and os.path.isdir(data_dir + 'labels/' + i))


def label_loader(label_path):
    """utils/dataloaders.py:81-83: float64 {0,1} change mask."""
    return read_png_gray(label_path + '/cm/' + 'cm.png') / 255


def get_train_val_metadata(data_dir, val_cities, patch_size, stride):
    """utils/dataloaders.py:51-78: [city, i, j] for every stride-grid origin whose patch fits the label raster."""
    cities = _cities(data_dir)
    train_cities = [c for c in cities if c not in set(val_cities)]
    train_metadata, val_metadata = [], []
    for group, meta in ((train_cities, train_metadata), (val_cities, val_metadata)):
        for city in group:
            h, w = read_png_gray(data_dir + 'labels/' + city + '/cm/cm.png').shape
            meta += [[city, i, j] for i, j in patch_origins(h, w, patch_size, stride)]
    return train_metadata, val_metadata


def city_stack_device(city, width, height, opt, device='cuda'):
    """The two dates of one city as a [2, C, height, width] float32 device tensor: every band file is decoded on the host,
    uploaded in its native resolution and type, normalised and resized on the device (bdn_ingest_band)."""
    out = torch.empty(2, len(opt.band_ids), height, width, dtype=torch.float32, device=device)
    st = _lib.stream_ptr()
    for d, sub in enumerate(('/imgs_1/*', '/imgs_2/*')):
        band_path = sorted(glob.glob(city + sub))[0][:-7]
        for c, bid in enumerate(opt.band_ids):
            band = read_tiff(band_path + bid + '.tif')
            is_f32 = band.dtype != np.uint16
            src = torch.from_numpy(np.ascontiguousarray(band.astype(np.float32) if is_f32 else band.view(np.int16))).to(device)
            call('bdn_ingest_band', 1 if is_f32 else 0, ptr(src), band.shape[0], band.shape[1],
                 float(opt.band_means[bid]), float(opt.band_stds[bid]), ptr(out[d, c]), height, width, st)
            src.record_stream(torch.cuda.current_stream(out.device))
    return out


def city_loader(city_meta):
    """utils/dataloaders.py:86-111, same argument list [city_dir, w, h, opt] (cv2.resize's dsize order) and the same
    numpy [2, C, h, w] float32 result; the arithmetic runs on the device."""
    city, w, h, opt = city_meta
    return city_stack_device(city, w, h, opt).cpu().numpy()


def full_onera_loader(data_dir, opt, device=None):
    """utils/dataloaders.py:115-145: {city: {'images': [2,C,H,W] float32, 'labels': uint8 [H,W]}}.  device=None keeps the
    reference's numpy images; device='cuda' leaves the stacks in HBM (what predict_scene and the tile gather read).
    Cities are processed one after another: the per-band work is on the GPU, not in a process pool."""
    dataset = {}
    for city in _cities(data_dir):
        label = label_loader(data_dir + 'labels/' + city)
        stack = city_stack_device(data_dir + 'images/' + city, label.shape[1], label.shape[0], opt)
        dataset[city] = {'images': stack if device is not None else stack.cpu().numpy(), 'labels': label.astype(np.uint8)}
    return dataset
