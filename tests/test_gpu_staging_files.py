"""GPU: file <-> HBM staging (csrc/staging.hip; ref imageop.h:52-97) and the product-side helpers of the pipelined default action:
parallel pread into the pinned ring, positioned writes behind a mark of the compute stream, the sample permutation kernel."""
import os
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dev(n_bytes):
    import torch
    return torch.zeros(n_bytes, dtype=torch.uint8, device="cuda")


def test_read_file_to_device_ranges_short_files_and_errors(ctx, tmp_path):
    """ReadFileContent(filePath, size, offset, total) with the buffer in HBM: whole file, a range across several 32 MiB slots at an
    odd offset, a range past the end (reads short, as fread does), an empty range, a missing file."""
    import torch
    rng = np.random.default_rng(3)
    n = (70 << 20) + 12345                                   # three ring slots, the last one ragged
    data = rng.integers(0, 256, n, dtype=np.uint8)
    path = str(tmp_path / "blob.bin")
    data.tofile(path)
    d = _dev(n + 64)
    got = ctx.read_file_to_device(path, d)
    ctx.sync()
    assert got == n and np.array_equal(d[:n].cpu().numpy(), data) and not d[n:].any()
    d.zero_()
    off, cnt = (33 << 20) + 7, (34 << 20) + 3
    got, t = ctx.read_file_to_device(path, d, offset=off, nbytes=cnt, want_ticket=True)
    ctx.stage_wait(t)
    ctx.sync()
    assert got == cnt and np.array_equal(d[:cnt].cpu().numpy(), data[off:off + cnt])
    got = ctx.read_file_to_device(path, d, offset=n - 1000, nbytes=5000)
    ctx.sync()
    assert got == 1000 and np.array_equal(d[:1000].cpu().numpy(), data[-1000:])
    assert ctx.read_file_to_device(path, d, offset=n + 10, nbytes=100) == 0
    with pytest.raises(ValueError, match="cannot open file"):                    # std::invalid_argument, imageop.h:55-57
        ctx.read_file_to_device(str(tmp_path / "missing.bin"), d)
    torch.cuda.synchronize()


@pytest.mark.parametrize("mode", ["pwrite", "mmap"])
def test_write_device_to_file_at_offsets_marks_and_append(ctx, tmp_path, mode, monkeypatch):
    """WriteBufferToFile with the buffer in HBM: whole file, append (stitcher.h:114-120), positioned blocks in any order behind marks of
    the compute stream, from two writer threads at once (two download lanes), both write routes."""
    import torch
    monkeypatch.setenv("OIP_FILE_WRITE", mode)
    rng = np.random.default_rng(5)
    n = (40 << 20) + 4096 + 18
    host = rng.integers(0, 256, n, dtype=np.uint8)
    d = torch.from_numpy(host).cuda()
    p = str(tmp_path / "out.bin")
    ctx.write_device_to_file(d, n, p)
    assert np.array_equal(np.fromfile(p, np.uint8), host)
    ctx.write_device_to_file(d, 1000, p, append=True)
    back = np.fromfile(p, np.uint8)
    assert back.size == n + 1000 and np.array_equal(back[n:], host[:1000]) and np.array_equal(back[:n], host)
    # blocks at their offsets, last block first, each behind the mark taken right after the kernel that produced it
    q = str(tmp_path / "blocks.bin")
    open(q, "wb").close()
    W, H = 4096, 6000
    src = torch.from_numpy(rng.integers(0, 4096, (H, W), dtype=np.uint16)).cuda()
    kb = np.stack([np.full(W, 1.0), np.full(W, 3.0)], 1)
    dkb = ctx.upload_kb(kb)
    dst = torch.zeros_like(src)
    cuts = [0, 1500, 3000, 4500, H]
    jobs = []
    for i in range(4):
        a, b = cuts[i], cuts[i + 1]
        ctx.rrc_u16(src[a:], dst[a:], W, b - a, dkb)
        jobs.append((a, b, ctx.compute_mark()))
    errs = []

    def writer(mine):
        try:
            for a, b, m in mine:
                ctx.write_device_to_file_at(dst, (b - a) * W * 2, q, a * W * 2, mark=m, byte_offset=a * W * 2)
        except Exception as ex:                                              # noqa: BLE001
            errs.append(ex)
    th = [threading.Thread(target=writer, args=(jobs[3:1:-1],)), threading.Thread(target=writer, args=(jobs[1::-1],))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    want = (src.cpu().numpy().astype(np.int64) + 3).astype(np.uint16)
    assert np.array_equal(np.fromfile(q, np.uint16).reshape(H, W), want)
    # a positioned write into the middle of an existing file leaves the rest alone
    ctx.write_device_to_file_at(d, 5000, p, 777)
    again = np.fromfile(p, np.uint8)
    assert again.size == n + 1000 and np.array_equal(again[777:5777], host[:5000]) and np.array_equal(again[5777:n], host[5777:n])
    assert np.array_equal(again[:777], host[:777])


def test_permute_u16x4_in_place(ctx):
    """the product's sample order on the device: cv::imwrite's (c2,c1,c0,c3) and GDAL band maps (imageop.h:529)"""
    import torch
    rng = np.random.default_rng(7)
    for npix in (1, 2, 7, 4096 * 33 + 1):
        img = rng.integers(0, 65536, (npix, 4), dtype=np.uint16)
        for order in ([2, 1, 0, 3], [0, 1, 2, 3], [3, 3, 0, 1]):
            d = torch.from_numpy(img.copy()).cuda()
            ctx.permute_u16x4(d, npix, order)
            ctx.sync()
            assert np.array_equal(d.cpu().numpy(), img[:, order]), (npix, order)
    with pytest.raises(ValueError, match="outside 0..3"):
        ctx.permute_u16x4(torch.zeros(8, dtype=torch.uint16, device="cuda"), 2, [0, 1, 2, 4])


@pytest.mark.parametrize("mode", ["", "pwrite"])
def test_file_sink_prepared_ahead_of_its_pixels(ctx, tmp_path, mode, monkeypatch):
    """oip_file_sink_*: a product file created, reserved and mapped before its pixels exist (the aligned image while the strip is
    still being read), then filled HBM -> pinned slot -> mapping by parallel copies.  A header written before the sink is opened
    stays, the payload lands at its offset behind a mark of the compute stream, bytes behind the payload (a TIFF directory)
    can be appended afterwards; OIP_FILE_WRITE=pwrite takes the fallback route with the same result."""
    import torch
    if mode:
        monkeypatch.setenv("OIP_FILE_WRITE", mode)
    rng = np.random.default_rng(11)
    W, H = 4096, 5000
    src = torch.from_numpy(rng.integers(0, 4096, (H, W), dtype=np.uint16)).cuda()
    dkb = ctx.upload_kb(np.stack([np.full(W, 1.0), np.full(W, 5.0)], 1))
    p = str(tmp_path / "product.bin")
    header = b"HEADER16" * 2
    with open(p, "wb") as f:
        f.write(header)
    nbytes = H * W * 2
    sink = ctx.file_sink_open(p, len(header) + nbytes)
    if not mode:
        assert os.path.getsize(p) == len(header) + nbytes                # reserved up front
    dst = torch.zeros_like(src)
    ctx.rrc_u16(src, dst, W, H, dkb)
    mark = ctx.compute_mark()
    half = (H // 2) * W * 2
    ctx.file_sink_write(sink, len(header) + half, dst, nbytes - half, mark=mark, byte_offset=half)      # second half first
    ctx.file_sink_write(sink, len(header), dst, half, mark=mark)
    with pytest.raises(ValueError):
        ctx.file_sink_write(sink, len(header) + 1, dst, nbytes)            # past the end of the sink
    ctx.file_sink_close(sink)
    with open(p, "ab") as f:
        f.write(b"TAIL")
    raw = open(p, "rb").read()
    assert raw[:len(header)] == header and raw[-4:] == b"TAIL" and len(raw) == len(header) + nbytes + 4
    got = np.frombuffer(raw, np.uint16, count=H * W, offset=len(header)).reshape(H, W)
    assert np.array_equal(got, (src.cpu().numpy().astype(np.int64) + 5).astype(np.uint16))
