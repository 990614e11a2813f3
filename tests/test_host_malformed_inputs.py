"""Malformed OBJ / MTL / PNG files against the host loaders built with AddressSanitizer + UndefinedBehaviorSanitizer
(CPU build of csrc/host/*.cpp behind the C API, tests/cpp/host_sanitizer_driver.cpp).

The reference panics on a file it cannot read or parse (src/core/asset.rs:72-75,118: `.expect`, `.unwrap`).  The
drop-in must not take the host process down: every malformed input has to come back as an error code (RT_ERR_IO /
RT_ERR_PARSE / RT_ERR_CAPACITY) or load as whatever tobj / png would have made of it -- never a crash, an overflow, an
unbounded allocation or undefined behaviour.  Corpus: hand-written cases (truncation, negative / zero / huge indices,
NaN and overflowing numbers, cyclic and missing mtllib, corrupt zlib streams, lying IHDR sizes, bad filter bytes,
truncated Adam7) plus seeded byte-level mutations of valid files."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

from conftest import ROOT

OK, INVALID, CAPACITY, IO, PARSE, OOM = 0, -1, -2, -6, -7, -8
DATA = os.path.join(ROOT, "tests", "data")


def png_bytes(w, h, ctype=6, depth=8, interlace=0, raw=None, idat=None, extra=b"", crc_ok=True):
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    if raw is None:
        stride = (w * ch * depth + 7) // 8
        rng = np.random.default_rng(w * 131 + h)
        raw = b"".join(bytes([rng.integers(0, 5)]) + rng.integers(0, 256, stride, dtype=np.uint8).tobytes() for _ in range(h))

    def chunk(t, d):
        c = zlib.crc32(t + d) & 0xffffffff
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", c if crc_ok else c ^ 1)
    z = zlib.compress(raw) if idat is None else idat
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, interlace)) + extra +
            chunk(b"IDAT", z) + chunk(b"IEND", b""))


def corpus(tmp):
    files = []   # (mode, path, allowed return codes)

    def add(mode, name, data, allowed):
        path = os.path.join(tmp, name)
        with open(path, "wb") as f:
            f.write(data if isinstance(data, bytes) else data.encode())
        files.append((mode, path, allowed))
        return path

    quirks = open(os.path.join(DATA, "quirks.obj"), "rb").read()
    mtl = open(os.path.join(DATA, "quirks.mtl"), "rb").read()
    add("obj", "quirks.mtl", mtl, None)  # (resource of the OBJ cases; not run on its own)
    files.pop()
    loads = {OK, IO, PARSE, CAPACITY}
    # ---- OBJ / MTL ----
    add("obj", "valid.obj", quirks, {OK})
    for cut in (1, 17, len(quirks) // 3, len(quirks) // 2, len(quirks) - 3):
        add("obj", f"truncated_{cut}.obj", quirks[:cut], loads)
    add("obj", "empty.obj", b"", loads)
    add("obj", "only_newlines.obj", b"\n\r\n\n", loads)
    add("obj", "binary_garbage.obj", bytes(np.random.default_rng(5).integers(0, 256, 4096, dtype=np.uint8)), loads)
    add("obj", "nul_bytes.obj", b"v 0 0 0\x00\x00\nv 1 0 0\nv 0 1 0\nf 1 2 3\x00\n", loads)
    add("obj", "index_zero.obj", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 0 1 2\n", loads)
    add("obj", "index_negative_out_of_range.obj", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf -1 -2 -9\n", loads)
    add("obj", "index_huge.obj", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 4294967297\nf 1 2 99999999999999999999\n", loads)
    add("obj", "index_before_any_vertex.obj", "f 1 2 3\nv 0 0 0\nv 1 0 0\nv 0 1 0\n", loads)
    add("obj", "vt_vn_out_of_range.obj", "v 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvn 0 0 1\nf 1/9/1 2/1/7 3/-5/-5\n", loads)
    add("obj", "slashes.obj", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1// 2/// 3/\nf / / /\nf 1/2/3/4 2 3\n", loads)
    add("obj", "nan_inf.obj", "v nan inf -inf\nv 1e999 -1e999 1e-999\nv 0x10 1,5 --3\nv 1 2\nv\nf 1 2 3\nf 1 2 4\n", loads)
    add("obj", "degenerate_faces.obj", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1\nf 1 2\nf 1 1 1\nf 1 2 3 3 3 3 3 3 3 3 3 3 3 3 3 3 3 3\n", loads)
    add("obj", "long_line.obj", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf " + "1 2 3 " * 20000 + "\n# " + "x" * 200000 + "\n", loads)
    add("obj", "long_token.obj", "v " + "9" * 5000 + " 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\ng " + "n" * 100000 + "\n", loads)
    add("obj", "many_groups.obj", "v 0 0 0\nv 1 0 0\nv 0 1 0\n" + "".join(f"g g{i}\nf 1 2 3\n" for i in range(450)), loads)
    add("obj", "missing_mtllib.obj", "mtllib does_not_exist.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nusemtl nope\nf 1 2 3\n", loads)
    add("obj", "mtllib_is_itself.obj", "mtllib mtllib_is_itself.obj\nv 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n", loads)
    add("obj", "cyc_a.mtl", "newmtl a\nKd 1 0 0\nmtllib cyc_b.mtl\n", None); files.pop()
    add("obj", "cyc_b.mtl", "newmtl b\nKd 0 1 0\nmtllib cyc_a.mtl\n", None); files.pop()
    add("obj", "cyclic_mtllib.obj", "mtllib cyc_a.mtl cyc_b.mtl cyc_a.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nusemtl a\nf 1 2 3\nusemtl b\nf 1 2 3\n", loads)
    add("obj", "bad.mtl", "Kd 1 1 1\nnewmtl\nnewmtl x\nKd nan\nKs 1e999 -3\nNs -50\nNi\nillum 99999999999\nKe a b c\nmap_Kd\nmap_Kd missing.png\n"
                          "newmtl y\nmap_Kd corrupt_zlib.png\nmap_Disp ../../../etc/passwd\n", None); files.pop()
    add("obj", "bad_mtl.obj", "mtllib bad.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nusemtl x\nf 1/1 2/1 3/1\nusemtl y\nf 1/1 2/1 3/1\n", loads)
    add("obj", "directory_as_file.obj", "mtllib .\nv 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n", loads)
    rng = np.random.default_rng(2024)
    for k in range(120):   # byte-level mutations of the valid file: flips, deletions, duplications
        b = bytearray(quirks)
        for _ in range(int(rng.integers(1, 12))):
            op = rng.integers(0, 3)
            i = int(rng.integers(0, len(b)))
            if op == 0:
                b[i] = int(rng.integers(0, 256))
            elif op == 1:
                del b[i:i + int(rng.integers(1, 9))]
            else:
                b[i:i] = b[i:i + int(rng.integers(1, 30))]
        add("obj", f"mut_{k}.obj", bytes(b), loads)
    # ---- PNG ----
    good = png_bytes(5, 3)
    png = {OK, IO}
    add("png", "valid_rgba.png", good, {OK})
    for ct, d in ((0, 1), (0, 16), (2, 8), (2, 16), (3, 2), (4, 8), (6, 16)):
        extra = b""
        if ct == 3:
            pl = bytes(range(12))
            extra = struct.pack(">I", 12) + b"PLTE" + pl + struct.pack(">I", zlib.crc32(b"PLTE" + pl) & 0xffffffff)
        add("png", f"valid_{ct}_{d}.png", png_bytes(7, 4, ct, d, extra=extra), {OK})
    add("png", "valid_adam7.png", png_bytes(9, 9, 6, 8, interlace=1,
                                             raw=b"".join(b"\0" + bytes(4 * pw) * 1 for pw, ph in ((2, 2), (1, 2), (3, 1), (2, 3), (5, 2), (4, 5), (9, 4)) for _ in range(ph))), {OK})
    for cut in (0, 7, 8, 20, 33, len(good) - 13, len(good) - 1):
        add("png", f"truncated_{cut}.png", good[:cut], png)
    add("png", "not_a_png.png", b"GIF89a" + bytes(64), {IO})
    add("png", "corrupt_zlib.png", png_bytes(5, 3, idat=b"\x78\x9c" + bytes(range(40))), {IO})
    add("png", "zlib_truncated.png", png_bytes(64, 64, idat=zlib.compress(bytes(64 * (64 * 4 + 1)))[:-9]), png)
    add("png", "idat_too_short.png", png_bytes(64, 64, raw=bytes(100)), {IO})
    add("png", "idat_empty.png", png_bytes(5, 3, idat=b""), {IO})
    add("png", "zero_size.png", png_bytes(0, 0, raw=b""), {IO})
    add("png", "huge_ihdr.png", png_bytes(0x7fffffff, 0x7fffffff, raw=bytes(64)), {IO})
    add("png", "huge_ihdr_16.png", png_bytes(0xffffffff, 0xffffffff, 6, 16, raw=bytes(64)), {IO})
    add("png", "wide_ihdr.png", png_bytes(0x40000000, 1, raw=bytes(64)), {IO})
    add("png", "zip_bomb.png", png_bytes(30000, 30000, idat=zlib.compress(bytes(50_000_000))), {IO})
    add("png", "bad_depth.png", png_bytes(5, 3, 6, 8).replace(struct.pack(">IIBB", 5, 3, 8, 6), struct.pack(">IIBB", 5, 3, 7, 6)), png)
    add("png", "bad_depth_for_type.png", png_bytes(5, 3, 2, 8).replace(struct.pack(">IIBB", 5, 3, 8, 2), struct.pack(">IIBB", 5, 3, 1, 2)), png)
    add("png", "bad_ctype.png", png_bytes(5, 3).replace(struct.pack(">IIBB", 5, 3, 8, 6), struct.pack(">IIBB", 5, 3, 8, 5)), {IO})
    add("png", "bad_interlace.png", png_bytes(5, 3, interlace=7), png)
    add("png", "bad_filter.png", png_bytes(5, 3, raw=b"".join(bytes([9]) + bytes(20) for _ in range(3))), png)
    add("png", "palette_missing.png", png_bytes(7, 4, 3, 8), png)
    add("png", "palette_short.png", png_bytes(7, 4, 3, 8, raw=b"".join(b"\0" + bytes([200] * 7) for _ in range(4)),
                                             extra=struct.pack(">I", 3) + b"PLTE" + b"abc" + struct.pack(">I", zlib.crc32(b"PLTEabc") & 0xffffffff)), png)
    add("png", "chunk_length_lies.png", good[:8] + struct.pack(">I", 0xfffffff0) + good[12:], {IO})
    add("png", "ihdr_short.png", b"\x89PNG\r\n\x1a\n" + struct.pack(">I", 4) + b"IHDR" + b"abcd" + bytes(4) + good[33:], png)
    add("png", "adam7_truncated.png", png_bytes(9, 9, interlace=1, raw=bytes(30)), {IO})
    for k in range(120):
        b = bytearray(good if k % 2 else png_bytes(16, 16, 2, 8, interlace=k % 4 == 0))
        for _ in range(int(rng.integers(1, 6))):
            op = rng.integers(0, 3)
            i = int(rng.integers(0, len(b)))
            if op == 0:
                b[i] = int(rng.integers(0, 256))
            elif op == 1:
                del b[i:i + int(rng.integers(1, 5))]
            else:
                b[i:i] = b[i:i + int(rng.integers(1, 9))]
        add("png", f"mut_{k}.png", bytes(b), png)
    return files


@pytest.mark.slow
def test_malformed_inputs_come_back_as_error_codes_under_asan_and_ubsan(tmp_path):
    from ray_tracer_2_amd.build import build_host_sanitizer_driver
    exe = build_host_sanitizer_driver()
    files = corpus(str(tmp_path))
    manifest = tmp_path / "manifest.txt"
    manifest.write_text("".join(f"{mode} {path}\n" for mode, path, _ in files))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:allocator_may_return_null=0:max_allocation_size_mb=2048",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    out = subprocess.run([exe, str(manifest)], capture_output=True, text=True, timeout=900, env=env, errors="replace")
    lines = out.stdout.splitlines()
    begun = [ln[6:] for ln in lines if ln.startswith("BEGIN ")]
    ended = [int(ln[4:]) for ln in lines if ln.startswith("END ")]
    assert out.returncode == 0 and len(ended) == len(files), \
        f"the loaders went down on {begun[len(ended)] if len(ended) < len(begun) else '?'}:\n{out.stderr[-4000:]}"
    bad = [(os.path.basename(p), rc, sorted(allowed)) for (mode, p, allowed), rc in zip(files, ended) if rc not in allowed]
    assert not bad, bad
    # the valid files do load, and most of the hand-written malformed ones are refused (not silently accepted)
    assert sum(rc != OK for (_, p, _), rc in zip(files, ended) if "mut_" not in p) >= 30
