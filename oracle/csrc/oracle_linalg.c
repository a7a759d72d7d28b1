/* ORACLE (test infrastructure only - never linked into or called by the product path).
 *
 * Plain-C restatement of the linear-algebra kernels the reference gets from PETSc for the adjoint solve
 * (un-vendored dependency, petsc4py>=3.11, reference setup.py:41; call sites reference
 * src/adjoint/DALinearEqn/DALinearEqn.C:28-339 createMLRKSP, :341-437 solveLinearEqn):
 *   - CSR mat-vec (MatMult of the assembled dRdWT),
 *   - ILU(k) by level of fill (PCILU + PCFactorSetLevels, DALinearEqn.C:268-299) with the
 *     non-zero pivot shift (PCFactorSetShiftType(MAT_SHIFT_NONZERO), :270-272),
 *   - forward/backward substitution.
 * PARITY UNPINNED: PETSc is not available here; the algorithms are the textbook ones (Saad, Iterative
 * Methods, Alg. 10.5 for ILU(p)); they are pinned against a dense LU on small matrices in tests/.
 * Also used, timed, as bench.py's cpu_baseline ("port").
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

void csr_spmv(long long n, const long long* rp, const int* ci, const double* v, const double* x, double* y) {
    for (long long i = 0; i < n; i++) {
        double s = 0.0;
        for (long long k = rp[i]; k < rp[i + 1]; k++) s += v[k] * x[ci[k]];
        y[i] = s;
    }
}

/* symbolic ILU(k): returns pattern with fill; rows sorted.  Caller frees with ilu_free. */
typedef struct {
    long long n, nnz;
    long long* rp;
    int* ci;
    int* lev;
    long long* diag;
} ilu_pat;

void ilu_free(ilu_pat* p) {
    if (!p) return;
    free(p->rp); free(p->ci); free(p->lev); free(p->diag); free(p);
}

static int cmp_int(const void* a, const void* b) { return (*(const int*)a > *(const int*)b) - (*(const int*)a < *(const int*)b); }

ilu_pat* ilu_symbolic(long long n, const long long* rp, const int* ci, int lfill) {
    ilu_pat* P = (ilu_pat*)calloc(1, sizeof(ilu_pat));
    long long cap = rp[n] * (lfill > 0 ? 3 : 1) + n + 16;
    P->n = n;
    P->rp = (long long*)malloc((n + 1) * sizeof(long long));
    P->ci = (int*)malloc(cap * sizeof(int));
    P->lev = (int*)malloc(cap * sizeof(int));
    P->diag = (long long*)malloc(n * sizeof(long long));
    int* wlev = (int*)malloc(n * sizeof(int)); /* level of column j in the current row, -1 = absent */
    int* list = (int*)malloc(n * sizeof(int));
    for (long long j = 0; j < n; j++) wlev[j] = -1;
    P->rp[0] = 0;
    for (long long i = 0; i < n; i++) {
        int cnt = 0, hasdiag = 0;
        for (long long k = rp[i]; k < rp[i + 1]; k++) {
            int j = ci[k];
            if (wlev[j] < 0) { wlev[j] = 0; list[cnt++] = j; }
            if (j == i) hasdiag = 1;
        }
        if (!hasdiag) { wlev[i] = 0; list[cnt++] = (int)i; }
        if (lfill > 0) {
            /* process pivots k < i in increasing order; the list grows, so re-sort lazily */
            qsort(list, cnt, sizeof(int), cmp_int);
            int pos = 0;
            while (pos < cnt && list[pos] < i) {
                int k = list[pos];
                int lk = wlev[k];
                if (lk < lfill + 1) {
                    /* add U-part of row k */
                    int added = 0;
                    for (long long q = P->diag[k] + 1; q < P->rp[k + 1]; q++) {
                        int j = P->ci[q];
                        int nl = lk + P->lev[q] + 1;
                        if (nl > lfill) continue;
                        if (wlev[j] < 0) { wlev[j] = nl; list[cnt++] = j; added = 1; }
                        else if (nl < wlev[j]) wlev[j] = nl;
                    }
                    if (added) qsort(list + pos + 1, cnt - pos - 1, sizeof(int), cmp_int);
                }
                pos++;
            }
        } else {
            qsort(list, cnt, sizeof(int), cmp_int);
        }
        if (P->rp[i] + cnt > cap) {
            cap = (P->rp[i] + cnt) * 2;
            P->ci = (int*)realloc(P->ci, cap * sizeof(int));
            P->lev = (int*)realloc(P->lev, cap * sizeof(int));
        }
        long long b = P->rp[i];
        for (int t = 0; t < cnt; t++) {
            int j = list[t];
            P->ci[b + t] = j;
            P->lev[b + t] = wlev[j];
            if (j == i) P->diag[i] = b + t;
            wlev[j] = -1;
        }
        P->rp[i + 1] = b + cnt;
    }
    P->nnz = P->rp[n];
    free(wlev); free(list);
    return P;
}

long long ilu_nnz(ilu_pat* p) { return p->nnz; }
void ilu_get(ilu_pat* p, long long* rp, int* ci, long long* diag) {
    memcpy(rp, p->rp, (p->n + 1) * sizeof(long long));
    memcpy(ci, p->ci, p->nnz * sizeof(int));
    memcpy(diag, p->diag, p->n * sizeof(long long));
}

/* numeric ILU on a given pattern (IKJ); L unit lower stored strictly-lower, U incl. diagonal.
 * shift: if |pivot| < tol the pivot is replaced by sign*shift (MAT_SHIFT_NONZERO analogue). */
int ilu_numeric(long long n, const long long* arp, const int* aci, const double* av, const long long* frp, const int* fci,
                const long long* fdiag, double* fv, double shift) {
    long long* where = (long long*)malloc(n * sizeof(long long));
    for (long long j = 0; j < n; j++) where[j] = -1;
    int nshift = 0;
    for (long long i = 0; i < n; i++) {
        for (long long q = frp[i]; q < frp[i + 1]; q++) { fv[q] = 0.0; where[fci[q]] = q; }
        for (long long k = arp[i]; k < arp[i + 1]; k++) {
            long long q = where[aci[k]];
            if (q >= 0) fv[q] = av[k];
        }
        for (long long q = frp[i]; q < fdiag[i]; q++) {
            int k = fci[q];
            double lik = fv[q] / fv[fdiag[k]];
            fv[q] = lik;
            if (lik == 0.0) continue;
            for (long long r = fdiag[k] + 1; r < frp[k + 1]; r++) {
                long long d = where[fci[r]];
                if (d >= 0) fv[d] -= lik * fv[r];
            }
        }
        double piv = fv[fdiag[i]];
        if (fabs(piv) < 1e-300 || piv != piv) { fv[fdiag[i]] = (piv < 0 ? -1.0 : 1.0) * shift; nshift++; }
        for (long long q = frp[i]; q < frp[i + 1]; q++) where[fci[q]] = -1;
    }
    free(where);
    return nshift;
}

void ilu_solve(long long n, const long long* frp, const int* fci, const long long* fdiag, const double* fv, const double* b, double* x) {
    for (long long i = 0; i < n; i++) {
        double s = b[i];
        for (long long q = frp[i]; q < fdiag[i]; q++) s -= fv[q] * x[fci[q]];
        x[i] = s;
    }
    for (long long i = n - 1; i >= 0; i--) {
        double s = x[i];
        for (long long q = fdiag[i] + 1; q < frp[i + 1]; q++) s -= fv[q] * x[fci[q]];
        x[i] = s / fv[fdiag[i]];
    }
}

/* y[0..m) = V[0..m)^T w ; V row-major m x n */
void multi_dot(long long n, int m, const double* V, const double* w, double* y) {
    for (int i = 0; i < m; i++) {
        const double* v = V + (long long)i * n;
        double s = 0.0;
        for (long long k = 0; k < n; k++) s += v[k] * w[k];
        y[i] = s;
    }
}
void multi_axpy(long long n, int m, const double* V, const double* h, double* w) {
    for (int i = 0; i < m; i++) {
        const double* v = V + (long long)i * n;
        double a = h[i];
        for (long long k = 0; k < n; k++) w[k] -= a * v[k];
    }
}
