/*
 * oracle/mirror.c -- CPU ORACLE (test infrastructure).
 *
 * regularize_method = "MIRROR" (generate_acados_solver.py:157, "Necessary to converge").
 * [UPSTREAM acados regularize_mirror]: W = V diag(e) V^T, e_i <- (|e_i| if |e_i| > eps else eps),
 * W <- V diag(e) V^T.  Computed here with a cyclic Jacobi eigen-iteration; only the reconstructed
 * matrix is exposed, so eigenvector sign/order ambiguity cannot leak.
 */
#include <math.h>
#include "tmpc_oracle.h"

void orc_mirror(double *W, int n, double eps)
{
    double A[8][8], V[8][8];
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            A[i][j] = 0.5 * (W[i * n + j] + W[j * n + i]);
            V[i][j] = (i == j) ? 1.0 : 0.0;
        }
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < n; i++) {
            diag += A[i][i] * A[i][i];
            for (int j = i + 1; j < n; j++) off += A[i][j] * A[i][j];
        }
        if (off <= 1e-32 * (diag + off) || off == 0.0) break;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) {
                double apq = A[p][q];
                if (apq == 0.0) continue;
                double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
                double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; k++) {        /* A <- A J */
                    double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq; A[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; k++) {        /* A <- J^T A */
                    double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk; A[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; k++) {
                    double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq;
                }
            }
    }
    double e[8];
    for (int i = 0; i < n; i++) {
        double ei = A[i][i];
        if (ei >= -eps && ei <= eps) ei = eps; else if (ei < 0.0) ei = -ei;
        e[i] = ei;
    }
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            double acc = 0.0;
            for (int k = 0; k < n; k++) acc += V[i][k] * e[k] * V[j][k];
            W[i * n + j] = acc;
        }
}
