"""The loggam table of the device rng.poisson (vkx_np_poisson_loggam_table: random_loggam restated on the host side of
vkit_amd/csrc/poisson.hip) against numpy's OWN random_loggam, linked from the static library numpy ships for its C API
(numpy/random/lib/libnpyrandom.a): bit for bit, for every argument the PTRS comparison can look up."""
import ctypes
import os
import shutil
import subprocess

import numpy as np
import pytest

SRC = r'''
#include <stdio.h>
#include <string.h>
double random_loggam(double x);
int main(void) {
    for (int x = 1; x <= 1024; x++) {
        double v = random_loggam((double)x);
        unsigned long long u;
        memcpy(&u, &v, 8);
        printf("%llx\n", u);
    }
    return 0;
}
'''


def test_loggam_table_is_numpys(tmp_path):
    from vkit_amd import _native as N
    lib = os.path.join(os.path.dirname(np.__file__), 'random', 'lib', 'libnpyrandom.a')
    if not os.path.exists(lib) or shutil.which('gcc') is None:
        pytest.skip('numpy ships no libnpyrandom.a here, or no gcc')
    (tmp_path / 'ref.c').write_text(SRC)
    exe = str(tmp_path / 'ref')
    subprocess.run(['gcc', '-O1', str(tmp_path / 'ref.c'), lib, '-lm', '-o', exe], check=True)
    ref = np.array([int(line, 16) for line in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()], dtype=np.uint64)
    out = np.zeros(1024)
    N.check(N.lib().vkx_np_poisson_loggam_table(out.ctypes.data_as(ctypes.c_void_p), 1024))
    np.testing.assert_array_equal(out.view(np.uint64), ref)
    assert out[0] == 0.0 and out[1] == 0.0 and abs(out[9] - 12.801827480081469) < 1e-12      # log(9!)
