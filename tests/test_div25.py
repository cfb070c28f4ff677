"""The batched matcher replaces `y / 25.0f` (hash cell of lvt_image_features_struct.cpp:71-83) by three IEEE operations
(k_hamming.hip: div_cell).  That is only legal if the quotient is bit-identical for every input; this test re-runs the
sweep on a sample (every 64th finite float of both signs, ~67M values) with gcc, contraction off like the device build."""
import os
import subprocess
import sys

SRC = r"""
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
static inline float div_cell(float y) {
    const float c = 0.04f;
    const float q0 = y * c;
    return fmaf(fmaf(-25.0f, q0, y), c, q0);
}
int main(void) {
    unsigned long long bad = 0, n = 0;
    for (uint32_t u = 0; u < 0x7F800000u; u += 64) {
        for (int sgn = 0; sgn < 2; sgn++) {
            const uint32_t w = u | ((uint32_t)sgn << 31);
            float y;
            memcpy(&y, &w, 4);
            const float a = y / 25.0f, b = div_cell(y);
            if (a != b) bad++; /* value equality: -0 / 25 may come out as +0, the floor is the same */
            n++;
        }
    }
    /* the coordinates the matcher really sees: every 1/64 px of a 4096-px axis */
    for (int i = -64 * 64; i < 4096 * 64; i++) {
        const float y = (float)i / 64.0f;
        if (floorf(y / 25.0f) != floorf(div_cell(y))) bad++;
        n++;
    }
    printf("%llu %llu\n", n, bad);
    return bad != 0;
}
"""


def test_div_cell_is_exact(tmp_path):
    src = tmp_path / "div25.c"
    src.write_text(SRC)
    exe = tmp_path / "div25"
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-o", str(exe), str(src), "-lm"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    n, bad = (int(v) for v in out.stdout.split())
    assert out.returncode == 0 and bad == 0, f"{bad} of {n} quotients differ"
    assert n > 60_000_000


def test_floor_of_the_float_quotient_is_the_exact_floor():
    """k_hamming.hip prunes a query's hash rows by their distance to the query: a feature of row ry is taken to lie in the slab
    25 ry <= y < 25 (ry + 1).  The row is floor(fl(y / 25.0f)); this checks that rounding the quotient never lifts it across an
    integer: for every k, the largest float below 25 k still divides to something below k (division is monotone, so
    floor(fl(y / 25)) >= k  <=>  y >= 25 k for every float y >= 0)."""
    import numpy as np
    k = np.arange(1, 4001, dtype=np.float32)
    edge = (k * np.float32(25.0)).astype(np.float32)                # exactly representable
    below = np.nextafter(edge, np.float32(0))
    assert (np.floor((below / np.float32(25.0)).astype(np.float32)) == k - 1).all()
    assert (np.floor((edge / np.float32(25.0)).astype(np.float32)) == k).all()
