"""Host build of a generated stage-function header (g++), for checking generated code on the CPU: the same header the
kernel compiles, wrapped in extern "C" entry points and loaded with ctypes.  Used by tests and by build.py's self-check."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

_WRAP = r'''
#include "%s"
extern "C" {
int gen_npar() { return tmpc_gen::NPAR; }
int gen_nh() { return tmpc_gen::NH; }
int gen_slack() { return tmpc_gen::SLACK; }
void gen_row_meta(int *src, int *sign, double *bound)
{ for (int k = 0; k < tmpc_gen::NH; k++) { src[k] = tmpc_gen::ROW_SRC[k]; sign[k] = tmpc_gen::ROW_SIGN[k]; bound[k] = tmpc_gen::ROW_BOUND[k]; } }
void gen_cost(const double *z, const double *p, double slack, double *val, double *g, double *H)
{ tmpc_gen::cost_full(z, p, 1, slack, val, g, H); double v2; tmpc_gen::cost_value(z, p, 1, slack, &v2); val[1] = v2; }
void gen_rows(const double *z, const double *p, double slack, double *h, double *D, double *Hr)
{
    tmpc_gen::rows(z, p, 1, slack, [&](int k, double hv, double gx, double gy, double gp, double hxx, double hxy, double hyy,
                                       double hxp, double hyp, double hpp) {
        h[k] = hv; D[3 * k] = gx; D[3 * k + 1] = gy; D[3 * k + 2] = gp;
        double *q = Hr + 6 * k; q[0] = hxx; q[1] = hxy; q[2] = hyy; q[3] = hxp; q[4] = hyp; q[5] = hpp;
    });
}
}
'''


class HostStageFunctions:
    def __init__(self, header_text):
        self._dir = tempfile.mkdtemp(prefix="tmpc_gen_")
        hpath = os.path.join(self._dir, "stage.h")
        with open(hpath, "w") as fh:
            fh.write(header_text)
        cpath = os.path.join(self._dir, "wrap.cpp")
        with open(cpath, "w") as fh:
            fh.write(_WRAP % hpath)
        so = os.path.join(self._dir, "libgen.so")
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-ffp-contract=off", "-o", so, cpath])
        self.lib = C.CDLL(so)
        self.npar, self.nh, self.slack = self.lib.gen_npar(), self.lib.gen_nh(), self.lib.gen_slack()
        src = np.zeros(max(self.nh, 1), np.int32); sgn = np.zeros(max(self.nh, 1), np.int32); bnd = np.zeros(max(self.nh, 1))
        self.lib.gen_row_meta(src.ctypes.data_as(C.c_void_p), sgn.ctypes.data_as(C.c_void_p), bnd.ctypes.data_as(C.c_void_p))
        self.row_src, self.row_sign, self.row_bound = src[:self.nh], sgn[:self.nh], bnd[:self.nh]

    @staticmethod
    def _p(a):
        return a.ctypes.data_as(C.c_void_p)

    def cost(self, z, p, slack=0.0):
        z = np.ascontiguousarray(z, float)[:7].copy(); p = np.ascontiguousarray(p, float)
        val = np.zeros(2); g = np.zeros(7); H = np.zeros(28)
        self.lib.gen_cost(self._p(z), self._p(p), C.c_double(slack), self._p(val), self._p(g), self._p(H))
        Hf = np.zeros((7, 7))
        for i in range(7):
            for j in range(i + 1):
                Hf[i, j] = Hf[j, i] = H[i * (i + 1) // 2 + j]
        assert val[0] == val[1] or abs(val[0] - val[1]) <= 1e-14 * max(1.0, abs(val[0]))
        return val[0], g, Hf

    def rows(self, z, p, slack=0.0):
        """Normalised rows g_k <= 0: values [NH], Jacobian w.r.t. (x, y, psi) [NH][3], Hessians [NH][3][3]."""
        z = np.ascontiguousarray(z, float)[:7].copy(); p = np.ascontiguousarray(p, float)
        h = np.zeros(self.nh); D = np.zeros((self.nh, 3)); Hr = np.zeros((self.nh, 6))
        self.lib.gen_rows(self._p(z), self._p(p), C.c_double(slack), self._p(h), self._p(D), self._p(Hr))
        H = np.zeros((self.nh, 3, 3))
        for k in range(self.nh):
            xx, xy, yy, xp, yp, pp = Hr[k]
            H[k] = [[xx, xy, xp], [xy, yy, yp], [xp, yp, pp]]
        return h, D, H
