"""The stripe and checkpoint geometry of the DP kernels (paramugsy_amd/csrc/dp_internal.hpp: full stripes, the narrow last stripes of
round 4, column groups, the words of the checkpoint workspace), compiled for the host and checked for the invariants the fill kernel
and the walk rely on -- without a GPU.  A test of the kernels' own code (tests/tools/dp_geometry_harness.cpp), not a product path."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def geo(tmp_path_factory):
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    out = str(tmp_path_factory.mktemp("geo") / "libdp_geometry.so")
    subprocess.run(["hipcc", "-O1", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-Wno-option-ignored", "-o", out,
                    os.path.join(ROOT, "tests", "tools", "dp_geometry_harness.cpp")], check=True, capture_output=True)
    h = C.CDLL(out)
    LL = C.c_longlong
    for name, args in (("geo_stripes", [LL, C.c_int, C.c_int]), ("geo_padded_cols", [LL, C.c_int, C.c_int]), ("geo_groups", [LL, C.c_int, C.c_int]),
                       ("geo_words", [LL, LL, C.c_int, C.c_int]), ("geo_bytes_written", [LL, LL, C.c_int, C.c_int]),
                       ("geo_cost", [LL, LL, C.c_int, C.c_int]), ("geo_col_word", [LL, LL, LL]), ("geo_row_word", [LL, LL, C.c_int, C.c_int, LL, LL]),
                       ("geo_nck", [LL])):
        getattr(h, name).argtypes = args
        getattr(h, name).restype = LL
    h.geo_stripe.argtypes = [LL, C.c_int, C.c_int, LL, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    h.geo_stripe_of_col.argtypes = [LL, C.c_int, C.c_int, LL, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    h.geo_group_lane0.argtypes = [LL, C.c_int, C.c_int, LL]
    h.geo_group_lanes.argtypes = [LL, C.c_int, C.c_int, LL]
    return h


def stripes_of(h, lb, Cc, tail):
    out = []
    jb, cs = C.c_int(), C.c_int()
    for s in range(h.geo_stripes(lb, Cc, tail)):
        h.geo_stripe(lb, Cc, tail, s, C.byref(jb), C.byref(cs))
        out.append((jb.value, cs.value))
    return out


def test_stripes_tile_the_columns_and_the_narrow_ones_come_last(geo):
    for Cc, tail in ((16, 0), (16, 1), (8, 0), (8, 1)):
        for lb in list(range(0, 2200)) + [4095, 4096, 4097, 4352, 4608, 4864, 5120, 9999, 10000, 16 << 20]:
            st = stripes_of(geo, lb, Cc, tail)
            at = 0
            for jb, cs in st:  # contiguous, 64 lanes x cs columns each
                assert jb == at and cs in ((16, 8, 4) if Cc == 16 else (8,))
                at += 64 * cs
            assert at == geo.geo_padded_cols(lb, Cc, tail) >= lb
            widths = [cs for _, cs in st]
            assert widths == sorted(widths, reverse=True)  # full stripes first, then narrower ones
            if not (tail and Cc == 16):
                assert all(cs == Cc for cs in widths) and at - lb < 64 * Cc
            else:
                assert at - lb < 256 or lb == 0  # the padding is less than a quarter of a full stripe
                assert sum(1 for cs in widths if cs != 16) <= 2
            if lb:
                jb, cs = C.c_int(), C.c_int()
                for j in {0, lb - 1, lb // 2, max(0, lb - 257), max(0, lb - 513)}:  # the stripe a column lies in
                    geo.geo_stripe_of_col(lb, Cc, tail, j, C.byref(jb), C.byref(cs))
                    assert (jb.value, cs.value) in st and jb.value <= j < jb.value + 64 * cs.value


def test_checkpoint_words_of_a_pair_are_disjoint_and_every_store_is_a_whole_aligned_line(geo):
    """Every column checkpoint (group, step) and every row checkpoint (m, column) of a pair has a slot of its own inside the words the
    planner allots the pair; what is left over is alignment (round 5): a pair's workspace is a multiple of 128 bytes, the 16 steps of
    a group that the fill kernel writes with one store are one aligned 128-byte line, the row checkpoints begin on a line and a
    lane's columns of one checkpoint are consecutive -- with 16 columns a lane exactly one aligned line."""
    R, W = geo.geo_ck_r(), geo.geo_ck_w()
    rng = np.random.default_rng(4)
    shapes = [(1, 1), (63, 64), (64, 1024), (65, 1025), (200, 300), (130, 700), (100, 1300), (77, 1800), (90, 2048), (150, 2100)]
    shapes += [(int(rng.integers(1, 260)), int(rng.integers(1, 2400))) for _ in range(12)]
    for Cc, tail in ((16, 0), (16, 1), (8, 0)):
        for la, lb in shapes:
            words = geo.geo_words(la, lb, Cc, tail)
            assert words % 32 == 0  # 128 bytes: the next pair's workspace begins on a line
            seen = np.zeros(words // 2, dtype=np.uint8)  # int2 slots
            steps, nck, groups, padded = la + 63, geo.geo_nck(la), geo.geo_groups(lb, Cc, tail), geo.geo_padded_cols(lb, Cc, tail)
            assert nck == (la + 63) // R and groups * W * Cc == padded
            for g in range(groups):
                w = np.array([geo.geo_col_word(la, g, t) for t in (0, 1, steps // 2, steps - 1)])
                assert np.all(w % 2 == 0) and np.all(w + 2 <= words)
                first = geo.geo_col_word(la, g, 0) // 2
                assert geo.geo_col_word(la, g, steps - 1) // 2 == first + steps - 1  # a group's steps are contiguous
                assert first % 16 == 0  # step 0, 16, 32, ... of every group: the start of a line
                seen[first:first + steps] += 1
            jb, cs = C.c_int(), C.c_int()
            for m in range(nck):
                for j in range(padded):
                    w = geo.geo_row_word(la, lb, Cc, tail, m, j)
                    assert w % 2 == 0 and w + 2 <= words
                    seen[w // 2] += 1
                    geo.geo_stripe_of_col(lb, Cc, tail, j, C.byref(jb), C.byref(cs))
                    c = (j - jb.value) % cs.value
                    if c == 0:
                        assert (w // 2) % cs.value == 0  # a lane's columns: cs consecutive slots, aligned to their own size
                    else:
                        assert w == geo.geo_row_word(la, lb, Cc, tail, m, j - 1) + 2
            assert np.all(seen <= 1), (Cc, tail, la, lb, int((seen > 1).sum()))
            used = int(seen.sum())
            assert used == groups * steps + nck * padded
            assert words // 2 - used <= groups * 15 + 15  # the padding: less than 16 entries a group, and the end
            assert geo.geo_bytes_written(la, lb, Cc, tail) == (groups * la * 2 + nck * padded * 2) * 4 <= words * 4


def test_groups_are_whole_lanes_of_one_stripe_and_the_cost_follows_the_widths(geo):
    W = geo.geo_ck_w()
    for Cc, tail in ((16, 0), (16, 1), (8, 0)):
        for lb in (1, 255, 256, 257, 511, 512, 513, 767, 768, 769, 1023, 1024, 1025, 1280, 1536, 1792, 2048, 2050, 3000, 4096, 10000):
            st = stripes_of(geo, lb, Cc, tail)
            for g in range(geo.geo_groups(lb, Cc, tail)):
                first_col = g * W * Cc
                jb, cs = next((jb, cs) for jb, cs in st if jb <= first_col < jb + 64 * cs)
                lanes, lane0 = geo.geo_group_lanes(lb, Cc, tail, g), geo.geo_group_lane0(lb, Cc, tail, g)
                assert lanes * cs == W * Cc and lane0 * cs == first_col - jb and lane0 % lanes == 0 and lane0 + lanes <= 64
            la = 300
            assert geo.geo_cost(la, lb, Cc, tail) == (la + 63) * sum(6 * cs + 5 for _, cs in st)
        # the narrow stripes never cost more than the full one they replace
        for lb in range(1, 3000, 7):
            assert geo.geo_cost(500, lb, 16, 1) <= geo.geo_cost(500, lb, 16, 0)
