"""VelodyneKittiBinReader::read restated (threecrate-io/src/lidar.rs:310-343): 16-byte little-endian
records x, y, z, intensity -> n x 3 float32; a size that is not a multiple of 16 is InvalidData.
The reader is host code of the C-ABI library (no GPU needed)."""
import numpy as np
import pytest

import threecrate_amd as tc


def test_read_kitti_bin_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    rec = rng.normal(0, 20, (1237, 4)).astype("<f4")
    rec[:, 3] = rng.random(1237)
    p = tmp_path / "000000.bin"
    rec.tofile(p)
    out = tc.read_kitti_bin(str(p))
    assert out.dtype == np.float32 and out.shape == (1237, 3)
    assert np.array_equal(out, rec[:, :3])          # bit-exact, intensity dropped


def test_read_kitti_bin_empty_and_bad_size(tmp_path):
    p = tmp_path / "empty.bin"
    p.write_bytes(b"")
    assert tc.read_kitti_bin(str(p)).shape == (0, 3)
    q = tmp_path / "bad.bin"
    q.write_bytes(b"\0" * 40)                        # not a multiple of 16 (lidar.rs:321-326)
    with pytest.raises(tc.InvalidData):
        tc.read_kitti_bin(str(q))
    with pytest.raises(tc.InvalidData):
        tc.read_kitti_bin(str(tmp_path / "missing.bin"))
