/* ORACLE (test infrastructure only - never linked into or called by the product path).
 *
 * All-core CPU restatement of the reference's adjoint Krylov solve, used (a) as bench.py's `cpu_baseline` (kind "port") AT THE
 * BENCH SIZE and (b) as the independent CPU solve of the psi-parity checks at 200 k cells (tests/, bench.py).
 * What it restates (reference src/adjoint/DALinearEqn/DALinearEqn.C:28-339 createMLRKSP, :341-437 solveLinearEqn; PETSc itself is
 * un-vendored, petsc4py>=3.11, reference setup.py:41 - PARITY UNPINNED, textbook algorithms):
 *   - MatMult of the assembled dRdWT: CSR SpMV, OpenMP, rows cut into equal-nnz chunks, arrays FIRST-TOUCHED by the thread
 *     that later streams them (NUMA placement);
 *   - PCILU with a reordering (DALinearEqn.C:238-299: ILU + jacMatReOrdering): ONE global scalar ILU(0) of dRdWTPC in a given
 *     unknown order (the caller passes the permutation), factorised and applied LEVEL-SCHEDULED (rows of one dependency level in
 *     parallel, one barrier per level), non-zero pivot shift (PCFactorSetShiftType(MAT_SHIFT_NONZERO), :270-272);
 *   - optional additive piecewise-constant coarse correction on one scalar cell field (the product's two-level form);
 *   - KSPGMRES, right preconditioning, classical Gram-Schmidt with one refinement pass (KSP_GMRES_CGS_REFINE_*), unpreconditioned
 *     residual norm, restart; threaded multi-dot / multi-axpy over row chunks.
 * Pinned in tests/test_oracle_cpu.py against the serial kernels of oracle_linalg.c and scipy's sparse direct solve.
 */
#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef long long ll;

typedef struct {
    int nt;
    ll n;
    /* operator (CSR, first-touched per row chunk) */
    ll* arp; int* aci; double* av; ll* acut;
    /* preconditioner: ILU(0) of the permuted PC matrix */
    int has_pc;
    ll* frp; int* fci; double* fv; ll* fdiag;
    int* perm; /* new -> old */
    int nlevL, nlevU;
    ll *lpL, *lpU;
    int *rowsL, *rowsU;
    int nshift;
    double *xp, *bp;
    /* alternative preconditioner: node-block ILU(0) (dense 8x8 blocks on a node pattern, level order = node order) */
    int has_bilu, nNodes, nLv;
    int* nodeUnk;   /* nNodes x 8, -1 = empty slot */
    int* nodeOut;   /* nNodes x 8: the unknown a slot writes (multi-block structures: -1 for overlap copies); NULL = nodeUnk */
    ll* bptr; int* bcol; ll* bdiag;
    double* bval;   /* 64 per block: L blocks (row-scaled by the pivot inverse), U blocks */
    double* invD;   /* 64 per node */
    int* lvlPtr;
    double *by, *bx; /* nNodes x 8 work vectors */
    /* coarse space */
    int has_coarse, nagg;
    ll coff, cN;
    int* agg;
    double *Einv, *cr, *cz;
    /* timers */
    double t_spmv, t_pc, t_orth, t_total;
    ll c_spmv, c_pc;
} okry;

static double wall(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

static void* xmalloc(size_t b) {
    void* p = NULL;
    if (posix_memalign(&p, 2u << 20, b ? b : 64)) return NULL;
    return p;
}

okry* okry_create(int nthreads) {
    okry* k = (okry*)calloc(1, sizeof(okry));
    k->nt = nthreads > 0 ? nthreads : omp_get_max_threads();
    return k;
}
int okry_threads(const okry* k) { return k->nt; }

static void free_op(okry* k) { free(k->arp); free(k->aci); free(k->av); free(k->acut); k->arp = NULL; k->aci = NULL; k->av = NULL; k->acut = NULL; }
static void free_pc(okry* k) {
    free(k->frp); free(k->fci); free(k->fv); free(k->fdiag); free(k->perm); free(k->lpL); free(k->lpU); free(k->rowsL); free(k->rowsU);
    free(k->xp); free(k->bp);
    k->frp = NULL; k->fci = NULL; k->fv = NULL; k->fdiag = NULL; k->perm = NULL; k->lpL = k->lpU = NULL; k->rowsL = k->rowsU = NULL; k->xp = k->bp = NULL;
    k->has_pc = 0;
}
static void free_bilu(okry* k) {
    free(k->nodeUnk); free(k->nodeOut); free(k->bptr); free(k->bcol); free(k->bdiag); free(k->bval); free(k->invD); free(k->lvlPtr); free(k->by); free(k->bx);
    k->nodeUnk = NULL; k->nodeOut = NULL; k->bptr = NULL; k->bcol = NULL; k->bdiag = NULL; k->bval = NULL; k->invD = NULL; k->lvlPtr = NULL; k->by = k->bx = NULL;
    k->has_bilu = 0;
}
static void free_coarse(okry* k) { free(k->agg); free(k->Einv); free(k->cr); free(k->cz); k->agg = NULL; k->Einv = k->cr = k->cz = NULL; k->has_coarse = 0; }
void okry_free(okry* k) {
    if (!k) return;
    free_op(k); free_pc(k); free_bilu(k); free_coarse(k);
    free(k);
}

/* ---- host STREAM triad a = b + s c (three arrays of nd doubles, parallel first touch): GB/s of 24 nd bytes per sweep */
double okry_stream_GBps(okry* k, ll nd, int reps) {
    double *a = (double*)xmalloc(nd * 8), *b = (double*)xmalloc(nd * 8), *c = (double*)xmalloc(nd * 8);
    if (!a || !b || !c) { free(a); free(b); free(c); return -1.0; }
#pragma omp parallel for schedule(static) num_threads(k->nt)
    for (ll i = 0; i < nd; i++) { a[i] = 0.0; b[i] = 1.0; c[i] = 2.0; }
    double best = 1e300;
    for (int r = 0; r < reps; r++) {
        double t0 = wall();
#pragma omp parallel for schedule(static) num_threads(k->nt)
        for (ll i = 0; i < nd; i++) a[i] = b[i] + 3.0 * c[i];
        double dt = wall() - t0;
        if (dt < best) best = dt;
    }
    double chk = a[nd / 2];
    free(a); free(b); free(c);
    return chk == 7.0 ? 24.0 * nd / best / 1e9 : -1.0;
}

/* ---- operator ------------------------------------------------------------------------------------------------------ */
int okry_set_operator(okry* k, ll n, const ll* rp, const int* ci, const double* v) {
    free_op(k);
    k->n = n;
    const int nt = k->nt;
    const ll nnz = rp[n];
    k->arp = (ll*)xmalloc((n + 1) * sizeof(ll));
    k->aci = (int*)xmalloc(nnz * sizeof(int));
    k->av = (double*)xmalloc(nnz * sizeof(double));
    k->acut = (ll*)malloc((nt + 1) * sizeof(ll));
    if (!k->arp || !k->aci || !k->av || !k->acut) return -1;
    /* equal-nnz row chunks */
    k->acut[0] = 0;
    for (int t = 1; t < nt; t++) {
        ll target = nnz / nt * t, lo = k->acut[t - 1], hi = n;
        while (lo < hi) { ll mid = (lo + hi) / 2; if (rp[mid] < target) lo = mid + 1; else hi = mid; }
        k->acut[t] = lo;
    }
    k->acut[nt] = n;
#pragma omp parallel num_threads(nt)
    {
        const int t = omp_get_thread_num();
        const ll a = k->acut[t], b = k->acut[t + 1];
        for (ll i = a; i < b; i++) k->arp[i] = rp[i];
        if (t == nt - 1) k->arp[n] = rp[n];
        memcpy(k->aci + rp[a], ci + rp[a], (size_t)(rp[b] - rp[a]) * sizeof(int));
        memcpy(k->av + rp[a], v + rp[a], (size_t)(rp[b] - rp[a]) * sizeof(double));
    }
    return 0;
}

void okry_spmv(okry* k, const double* x, double* y) {
    const double t0 = wall();
#pragma omp parallel num_threads(k->nt)
    {
        const int t = omp_get_thread_num();
        const ll a = k->acut[t], b = k->acut[t + 1];
        const ll* rp = k->arp; const int* ci = k->aci; const double* v = k->av;
        for (ll i = a; i < b; i++) {
            double s = 0.0;
            for (ll q = rp[i]; q < rp[i + 1]; q++) s += v[q] * x[ci[q]];
            y[i] = s;
        }
    }
    k->t_spmv += wall() - t0;
    k->c_spmv++;
}

/* ---- ILU(0), level-scheduled ------------------------------------------------------------------------------------------ */
static void sort_row(int* c, double* v, ll len) {
    for (ll a = 1; a < len; a++) { /* rows are short (<= a few hundred) and nearly sorted after the permutation of a cell-ordered pattern */
        int cc = c[a]; double vv = v[a]; ll b = a - 1;
        while (b >= 0 && c[b] > cc) { c[b + 1] = c[b]; v[b + 1] = v[b]; b--; }
        c[b + 1] = cc; v[b + 1] = vv;
    }
}

/* perm: new -> old (NULL = identity).  Returns the number of shifted pivots, < 0 on error. */
int okry_set_pc_ilu0(okry* k, ll n, const ll* rp, const int* ci, const double* v, const int* perm, double shift) {
    free_pc(k);
    if (k->n && k->n != n) return -2;
    k->n = n;
    const int nt = k->nt;
    int* inv = (int*)xmalloc(n * sizeof(int));
    k->perm = (int*)xmalloc(n * sizeof(int));
    k->frp = (ll*)xmalloc((n + 1) * sizeof(ll));
    k->fdiag = (ll*)xmalloc(n * sizeof(ll));
    k->xp = (double*)xmalloc(n * 8); k->bp = (double*)xmalloc(n * 8);
    if (!inv || !k->perm || !k->frp || !k->fdiag || !k->xp || !k->bp) return -1;
#pragma omp parallel for schedule(static) num_threads(nt)
    for (ll i = 0; i < n; i++) { k->perm[i] = perm ? perm[i] : (int)i; }
#pragma omp parallel for schedule(static) num_threads(nt)
    for (ll i = 0; i < n; i++) inv[k->perm[i]] = (int)i;
    /* row lengths of the permuted matrix (+1 if the diagonal is structurally missing) */
    k->frp[0] = 0;
    for (ll i = 0; i < n; i++) {
        const ll o = k->perm[i];
        ll len = rp[o + 1] - rp[o];
        int has = 0;
        for (ll q = rp[o]; q < rp[o + 1]; q++) if (ci[q] == o) { has = 1; break; }
        k->frp[i + 1] = k->frp[i] + len + (has ? 0 : 1);
    }
    const ll nnz = k->frp[n];
    k->fci = (int*)xmalloc(nnz * sizeof(int));
    k->fv = (double*)xmalloc(nnz * sizeof(double));
    if (!k->fci || !k->fv) return -1;
#pragma omp parallel for schedule(static) num_threads(nt)
    for (ll i = 0; i < n; i++) {
        const ll o = k->perm[i];
        ll w = k->frp[i];
        int has = 0;
        for (ll q = rp[o]; q < rp[o + 1]; q++) { k->fci[w] = inv[ci[q]]; k->fv[w] = v[q]; if (ci[q] == o) has = 1; w++; }
        if (!has) { k->fci[w] = (int)i; k->fv[w] = 0.0; w++; }
        sort_row(k->fci + k->frp[i], k->fv + k->frp[i], w - k->frp[i]);
        for (ll q = k->frp[i]; q < w; q++) if (k->fci[q] == i) k->fdiag[i] = q;
    }
    free(inv);
    /* dependency levels of the forward (L) and backward (U) sweeps */
    int* lev = (int*)xmalloc(n * sizeof(int));
    int nl = 0;
    for (ll i = 0; i < n; i++) {
        int l = 0;
        for (ll q = k->frp[i]; q < k->fdiag[i]; q++) { int lk = lev[k->fci[q]] + 1; if (lk > l) l = lk; }
        lev[i] = l;
        if (l + 1 > nl) nl = l + 1;
    }
    k->nlevL = nl;
    k->lpL = (ll*)calloc(nl + 1, sizeof(ll));
    k->rowsL = (int*)xmalloc(n * sizeof(int));
    for (ll i = 0; i < n; i++) k->lpL[lev[i] + 1]++;
    for (int l = 0; l < nl; l++) k->lpL[l + 1] += k->lpL[l];
    {
        ll* pos = (ll*)malloc(nl * sizeof(ll));
        memcpy(pos, k->lpL, nl * sizeof(ll));
        for (ll i = 0; i < n; i++) k->rowsL[pos[lev[i]]++] = (int)i;
        free(pos);
    }
    nl = 0;
    for (ll i = n - 1; i >= 0; i--) {
        int l = 0;
        for (ll q = k->fdiag[i] + 1; q < k->frp[i + 1]; q++) { int lk = lev[k->fci[q]] + 1; if (lk > l) l = lk; }
        lev[i] = l;
        if (l + 1 > nl) nl = l + 1;
    }
    k->nlevU = nl;
    k->lpU = (ll*)calloc(nl + 1, sizeof(ll));
    k->rowsU = (int*)xmalloc(n * sizeof(int));
    for (ll i = 0; i < n; i++) k->lpU[lev[i] + 1]++;
    for (int l = 0; l < nl; l++) k->lpU[l + 1] += k->lpU[l];
    {
        ll* pos = (ll*)malloc(nl * sizeof(ll));
        memcpy(pos, k->lpU, nl * sizeof(ll));
        for (ll i = 0; i < n; i++) k->rowsU[pos[lev[i]]++] = (int)i;
        free(pos);
    }
    free(lev);
    /* numeric factorisation, IKJ, rows of one L level in parallel; sparse row updates by two-pointer merges (sorted rows) */
    int nshift = 0;
    const ll* frp = k->frp; const int* fci = k->fci; double* fv = k->fv; const ll* fdiag = k->fdiag;
#pragma omp parallel num_threads(nt) reduction(+ : nshift)
    {
        for (int l = 0; l < k->nlevL; l++) {
#pragma omp for schedule(dynamic, 64)
            for (ll r = k->lpL[l]; r < k->lpL[l + 1]; r++) {
                const ll i = k->rowsL[r];
                const ll ie = frp[i + 1];
                for (ll q = frp[i]; q < fdiag[i]; q++) {
                    const int kk = fci[q];
                    const double lik = fv[q] / fv[fdiag[kk]];
                    fv[q] = lik;
                    if (lik == 0.0) continue;
                    ll a = q + 1, b = fdiag[kk] + 1;
                    const ll be = frp[kk + 1];
                    while (a < ie && b < be) {
                        const int ca = fci[a], cb = fci[b];
                        if (ca == cb) { fv[a] -= lik * fv[b]; a++; b++; }
                        else if (ca < cb) a++;
                        else b++;
                    }
                }
                const double piv = fv[fdiag[i]];
                if (fabs(piv) < 1e-300 || piv != piv) { fv[fdiag[i]] = (piv < 0 ? -1.0 : 1.0) * shift; nshift++; }
            }
        }
    }
    k->nshift = nshift;
    k->has_pc = 1;
    return nshift;
}
int okry_pc_levels(const okry* k, int* nlevL, int* nlevU) { if (nlevL) *nlevL = k->nlevL; if (nlevU) *nlevU = k->nlevU; return k->has_pc; }
ll okry_pc_nnz(const okry* k) { return k->has_pc ? k->frp[k->n] : 0; }

/* ---- node-block ILU(0): the product's default preconditioner (csrc/das_bilu.hpp) restated for the host -------------------------------
 * nodes of <= 8 unknowns (nodeUnk, -1 = empty slot -> identity), block pattern bptr / bcol over nodes IN PROCESSING ORDER (sorted
 * columns, symmetric pattern), levels lvlPtr (nodes of one level are mutually independent and contiguous).  Block IKJ factorisation with
 * dense 8x8 blocks, pivot blocks inverted by Gauss-Jordan with row pivoting and a non-zero pivot shift; level-parallel sweeps. */
static int inv8(double* a, double* out, double shift) { /* a is destroyed */
    int nshift = 0;
    for (int i = 0; i < 64; i++) out[i] = (i / 8 == i % 8) ? 1.0 : 0.0;
    for (int c = 0; c < 8; c++) {
        int pr = c; double best = fabs(a[c * 8 + c]);
        for (int r = c + 1; r < 8; r++) if (fabs(a[r * 8 + c]) > best) { best = fabs(a[r * 8 + c]); pr = r; }
        if (pr != c) for (int q = 0; q < 8; q++) { double t = a[c * 8 + q]; a[c * 8 + q] = a[pr * 8 + q]; a[pr * 8 + q] = t; t = out[c * 8 + q]; out[c * 8 + q] = out[pr * 8 + q]; out[pr * 8 + q] = t; }
        double piv = a[c * 8 + c];
        if (fabs(piv) < 1e-300 || piv != piv) { piv = (piv < 0 ? -1.0 : 1.0) * shift; a[c * 8 + c] = piv; nshift++; }
        const double ip = 1.0 / piv;
        for (int q = 0; q < 8; q++) { a[c * 8 + q] *= ip; out[c * 8 + q] *= ip; }
        for (int r = 0; r < 8; r++) {
            if (r == c) continue;
            const double f = a[r * 8 + c];
            if (f == 0.0) continue;
            for (int q = 0; q < 8; q++) { a[r * 8 + q] -= f * a[c * 8 + q]; out[r * 8 + q] -= f * out[c * 8 + q]; }
        }
    }
    return nshift;
}
static inline void mm8(const double* a, const double* b, double* c) { /* c = a b */
    for (int i = 0; i < 8; i++)
        for (int j = 0; j < 8; j++) {
            double s = 0.0;
            for (int q = 0; q < 8; q++) s += a[i * 8 + q] * b[q * 8 + j];
            c[i * 8 + j] = s;
        }
}
int okry_set_pc_bilu2(okry* k, ll n, const ll* rp, const int* ci, const double* v, int nNodes, const int* nodeUnk, const int* nodeOut, const ll* bptr,
                      const int* bcol, int nLv, const int* lvlPtr, double shift);
int okry_set_pc_bilu(okry* k, ll n, const ll* rp, const int* ci, const double* v, int nNodes, const int* nodeUnk, const ll* bptr, const int* bcol,
                     int nLv, const int* lvlPtr, double shift) {
    return okry_set_pc_bilu2(k, n, rp, ci, v, nNodes, nodeUnk, NULL, bptr, bcol, nLv, lvlPtr, shift);
}
/* nodeOut != NULL: a multi-block structure (amd.pcSubdomains of the product): an unknown of an overlap ring sits in one node per block that
 * reaches it, the node patterns of the blocks are disjoint; a matrix entry (u, v) goes to the block (I, J) of every node I holding u and the
 * node J holding v INSIDE I's pattern; only the slot with nodeOut == u writes the solution */
int okry_set_pc_bilu2(okry* k, ll n, const ll* rp, const int* ci, const double* v, int nNodes, const int* nodeUnk, const int* nodeOut, const ll* bptr,
                      const int* bcol, int nLv, const int* lvlPtr, double shift) {
    free_bilu(k);
    if (k->n && k->n != n) return -2;
    k->n = n;
    const int nt = k->nt;
    const ll nB = bptr[nNodes];
    k->nNodes = nNodes; k->nLv = nLv;
    k->nodeUnk = (int*)xmalloc((size_t)nNodes * 8 * sizeof(int));
    k->bptr = (ll*)xmalloc((nNodes + 1) * sizeof(ll));
    k->bcol = (int*)xmalloc(nB * sizeof(int));
    k->bdiag = (ll*)xmalloc(nNodes * sizeof(ll));
    k->bval = (double*)xmalloc((size_t)nB * 64 * 8);
    k->invD = (double*)xmalloc((size_t)nNodes * 64 * 8);
    k->lvlPtr = (int*)malloc((nLv + 1) * sizeof(int));
    k->by = (double*)xmalloc((size_t)nNodes * 8 * 8); k->bx = (double*)xmalloc((size_t)nNodes * 8 * 8);
    /* slots holding an unknown: chained lists (head per unknown, next per slot) - one entry per unknown, or one per block for overlap unknowns */
    ll* unkHead = (ll*)xmalloc(n * sizeof(ll));
    ll* slotNext = (ll*)xmalloc((size_t)nNodes * 8 * sizeof(ll));
    if (nodeOut) { k->nodeOut = (int*)xmalloc((size_t)nNodes * 8 * sizeof(int)); if (!k->nodeOut) return -1; memcpy(k->nodeOut, nodeOut, (size_t)nNodes * 8 * sizeof(int)); }
    if (!k->nodeUnk || !k->bptr || !k->bcol || !k->bdiag || !k->bval || !k->invD || !k->lvlPtr || !k->by || !k->bx || !unkHead || !slotNext) return -1;
    memcpy(k->lvlPtr, lvlPtr, (nLv + 1) * sizeof(int));
#pragma omp parallel for schedule(static) num_threads(nt)
    for (ll u = 0; u < n; u++) unkHead[u] = -1;
    for (ll sl = 0; sl < (ll)nNodes * 8; sl++) {
        const int u = nodeUnk[sl];
        slotNext[sl] = -1;
        if (u >= 0) { slotNext[sl] = unkHead[u]; unkHead[u] = sl; }
    }
#pragma omp parallel for schedule(static) num_threads(nt)
    for (int I = 0; I < nNodes; I++) {
        k->bptr[I] = bptr[I];
        if (I == nNodes - 1) k->bptr[nNodes] = bptr[nNodes];
        for (int r = 0; r < 8; r++) k->nodeUnk[(size_t)I * 8 + r] = nodeUnk[(size_t)I * 8 + r];
        k->bdiag[I] = -1;
        for (ll e = bptr[I]; e < bptr[I + 1]; e++) { k->bcol[e] = bcol[e]; if (bcol[e] == I) k->bdiag[I] = e; memset(k->bval + (size_t)e * 64, 0, 64 * 8); }
    }
    for (int I = 0; I < nNodes; I++) if (k->bdiag[I] < 0) { free(unkHead); free(slotNext); return -3; }
    /* scatter the scalar entries into the dense blocks (entries outside the node pattern are dropped by construction) */
#pragma omp parallel for schedule(dynamic, 256) num_threads(nt)
    for (int I = 0; I < nNodes; I++) {
        for (int r = 0; r < 8; r++) {
            const int u = k->nodeUnk[(size_t)I * 8 + r];
            if (u < 0) { k->bval[(size_t)k->bdiag[I] * 64 + r * 8 + r] = 1.0; continue; }  /* empty slot: identity row */
            for (ll q = rp[u]; q < rp[u + 1]; q++) {
                for (ll sl = unkHead[ci[q]]; sl >= 0; sl = slotNext[sl]) {  /* the node holding the column unknown inside I's pattern (at most one) */
                    const int J = (int)(sl >> 3);
                    ll lo = k->bptr[I], hi = k->bptr[I + 1] - 1, e = -1;
                    while (lo <= hi) { const ll mid = (lo + hi) >> 1; const int cm = k->bcol[mid]; if (cm == J) { e = mid; break; } if (cm < J) lo = mid + 1; else hi = mid - 1; }
                    if (e >= 0) { k->bval[(size_t)e * 64 + r * 8 + (int)(sl & 7)] = v[q]; break; }
                }
            }
        }
    }
    free(unkHead); free(slotNext);
    /* block IKJ, the nodes of one level in parallel */
    int nshift = 0;
#pragma omp parallel num_threads(nt) reduction(+ : nshift)
    {
        double tmp[64], lik[64];
        for (int l = 0; l < nLv; l++) {
#pragma omp for schedule(dynamic, 8)
            for (int I = lvlPtr[l]; I < lvlPtr[l + 1]; I++) {
                const ll ie = k->bptr[I + 1];
                for (ll q = k->bptr[I]; q < k->bdiag[I]; q++) {
                    const int K = k->bcol[q];
                    mm8(k->bval + (size_t)q * 64, k->invD + (size_t)K * 64, lik);  /* L_IK = A_IK D_K^-1 */
                    memcpy(k->bval + (size_t)q * 64, lik, 64 * 8);
                    ll a = q + 1, b = k->bdiag[K] + 1;
                    const ll be = k->bptr[K + 1];
                    while (a < ie && b < be) {
                        const int ca = k->bcol[a], cb = k->bcol[b];
                        if (ca == cb) {
                            mm8(lik, k->bval + (size_t)b * 64, tmp);
                            double* d = k->bval + (size_t)a * 64;
                            for (int t = 0; t < 64; t++) d[t] -= tmp[t];
                            a++; b++;
                        } else if (ca < cb) a++;
                        else b++;
                    }
                }
                memcpy(tmp, k->bval + (size_t)k->bdiag[I] * 64, 64 * 8);
                nshift += inv8(tmp, k->invD + (size_t)I * 64, shift);
            }
        }
    }
    k->nshift = nshift;
    k->has_bilu = 1;
    return nshift;
}
static void bilu_apply(okry* k, const double* b, double* x) {
    const int nN = k->nNodes;
    double* y = k->by; double* z = k->bx;
#pragma omp parallel num_threads(k->nt)
    {
        for (int l = 0; l < k->nLv; l++) {
#pragma omp for schedule(static)
            for (int I = k->lvlPtr[l]; I < k->lvlPtr[l + 1]; I++) {
                double s[8];
                for (int r = 0; r < 8; r++) { const int u = k->nodeUnk[(size_t)I * 8 + r]; s[r] = u >= 0 ? b[u] : 0.0; }
                for (ll q = k->bptr[I]; q < k->bdiag[I]; q++) {
                    const double* L = k->bval + (size_t)q * 64; const double* yk = y + (size_t)k->bcol[q] * 8;
                    for (int r = 0; r < 8; r++) { double a = 0.0; for (int c = 0; c < 8; c++) a += L[r * 8 + c] * yk[c]; s[r] -= a; }
                }
                for (int r = 0; r < 8; r++) y[(size_t)I * 8 + r] = s[r];
            }
        }
        for (int l = k->nLv - 1; l >= 0; l--) {
#pragma omp for schedule(static)
            for (int I = k->lvlPtr[l]; I < k->lvlPtr[l + 1]; I++) {
                double s[8];
                for (int r = 0; r < 8; r++) s[r] = y[(size_t)I * 8 + r];
                for (ll q = k->bdiag[I] + 1; q < k->bptr[I + 1]; q++) {
                    const double* U = k->bval + (size_t)q * 64; const double* zj = z + (size_t)k->bcol[q] * 8;
                    for (int r = 0; r < 8; r++) { double a = 0.0; for (int c = 0; c < 8; c++) a += U[r * 8 + c] * zj[c]; s[r] -= a; }
                }
                const double* D = k->invD + (size_t)I * 64;
                for (int r = 0; r < 8; r++) { double a = 0.0; for (int c = 0; c < 8; c++) a += D[r * 8 + c] * s[c]; z[(size_t)I * 8 + r] = a; }
            }
        }
#pragma omp for schedule(static)
        for (int I = 0; I < nN; I++)
            for (int r = 0; r < 8; r++) { const int u = (k->nodeOut ? k->nodeOut : k->nodeUnk)[(size_t)I * 8 + r]; if (u >= 0) x[u] = z[(size_t)I * 8 + r]; }
    }
}
ll okry_bilu_blocks(const okry* k) { return k->has_bilu ? k->bptr[k->nNodes] : 0; }

/* coarse space on the scalar cell field at offset `off` (N cells): Einv = (Z^T P_ff Z)^-1 given by the caller */
int okry_set_coarse(okry* k, ll off, ll N, const int* agg, int nagg, const double* Einv) {
    free_coarse(k);
    if (nagg <= 0) return 0;
    k->coff = off; k->cN = N; k->nagg = nagg;
    k->agg = (int*)malloc(N * sizeof(int));
    k->Einv = (double*)malloc((size_t)nagg * nagg * 8);
    k->cr = (double*)malloc(nagg * 8); k->cz = (double*)malloc(nagg * 8);
    memcpy(k->agg, agg, N * sizeof(int));
    memcpy(k->Einv, Einv, (size_t)nagg * nagg * 8);
    k->has_coarse = 1;
    return 0;
}
/* E = Z^T P_ff Z of a CSR matrix (row-major nagg x nagg, caller-zeroed) */
void okry_coarse_operator(ll off, ll N, const int* agg, int nagg, const ll* rp, const int* ci, const double* v, double* E) {
    for (ll i = 0; i < N; i++) {
        const ll r = off + i;
        for (ll q = rp[r]; q < rp[r + 1]; q++) {
            const ll j = (ll)ci[q] - off;
            if (j >= 0 && j < N) E[(size_t)agg[i] * nagg + agg[j]] += v[q];
        }
    }
}

/* x = M^-1 b */
void okry_pc(okry* k, const double* b, double* x) {
    const double t0 = wall();
    const ll n = k->n;
    if (k->has_bilu) { bilu_apply(k, b, x); goto coarse; }
    if (!k->has_pc) { memcpy(x, b, n * 8); return; }
    {
    const ll* frp = k->frp; const int* fci = k->fci; const double* fv = k->fv; const ll* fdiag = k->fdiag;
    double* y = k->xp; double* bp = k->bp;
    const int* perm = k->perm;
#pragma omp parallel num_threads(k->nt)
    {
#pragma omp for schedule(static)
        for (ll i = 0; i < n; i++) bp[i] = b[perm[i]];
        for (int l = 0; l < k->nlevL; l++) {
#pragma omp for schedule(static)
            for (ll r = k->lpL[l]; r < k->lpL[l + 1]; r++) {
                const ll i = k->rowsL[r];
                double s = bp[i];
                for (ll q = frp[i]; q < fdiag[i]; q++) s -= fv[q] * y[fci[q]];
                y[i] = s;
            }
        }
        for (int l = 0; l < k->nlevU; l++) {
#pragma omp for schedule(static)
            for (ll r = k->lpU[l]; r < k->lpU[l + 1]; r++) {
                const ll i = k->rowsU[r];
                double s = y[i];
                for (ll q = fdiag[i] + 1; q < frp[i + 1]; q++) s -= fv[q] * y[fci[q]];
                y[i] = s / fv[fdiag[i]];
            }
        }
#pragma omp for schedule(static)
        for (ll i = 0; i < n; i++) x[perm[i]] = y[i];
    }
    }
coarse:
    if (k->has_coarse) {
        const int na = k->nagg;
        memset(k->cr, 0, na * 8);
        const double* bf = b + k->coff;
        for (ll i = 0; i < k->cN; i++) k->cr[k->agg[i]] += bf[i];
#pragma omp parallel for schedule(static) num_threads(k->nt)
        for (int a = 0; a < na; a++) {
            double s = 0.0;
            const double* e = k->Einv + (size_t)a * na;
            for (int c = 0; c < na; c++) s += e[c] * k->cr[c];
            k->cz[a] = s;
        }
        double* xf = x + k->coff;
#pragma omp parallel for schedule(static) num_threads(k->nt)
        for (ll i = 0; i < k->cN; i++) xf[i] += k->cz[k->agg[i]];
    }
    k->t_pc += wall() - t0;
    k->c_pc++;
}

/* ---- GMRES ------------------------------------------------------------------------------------------------------------ */
/* h[0..m) = V^T w and w -= V h, vectors cut into nt contiguous chunks (one per thread, the chunk of w stays in cache) */
static void multi_dot(const okry* k, double** V, int m, const double* w, double* h, double* part) {
    const int nt = k->nt;
    const ll n = k->n;
#pragma omp parallel num_threads(nt)
    {
        const int t = omp_get_thread_num();
        const ll a = n * t / nt, b = n * (t + 1) / nt;
        double* p = part + (size_t)t * m;
        for (int i = 0; i < m; i++) {
            const double* v = V[i];
            double s = 0.0;
            for (ll q = a; q < b; q++) s += v[q] * w[q];
            p[i] = s;
        }
    }
    for (int i = 0; i < m; i++) {
        double s = 0.0;
        for (int t = 0; t < nt; t++) s += part[(size_t)t * m + i];
        h[i] = s;
    }
}
static void multi_axpy(const okry* k, double** V, int m, const double* h, double* w) {
    const int nt = k->nt;
    const ll n = k->n;
#pragma omp parallel num_threads(nt)
    {
        const int t = omp_get_thread_num();
        const ll a = n * t / nt, b = n * (t + 1) / nt;
        for (int i = 0; i < m; i++) {
            const double* v = V[i];
            const double c = h[i];
            for (ll q = a; q < b; q++) w[q] -= c * v[q];
        }
    }
}
static double* vec_alloc(const okry* k) {
    double* v = (double*)xmalloc(k->n * 8);
    if (!v) return NULL;
    const ll n = k->n; const int nt = k->nt;
#pragma omp parallel num_threads(nt)
    {
        const int t = omp_get_thread_num();
        const ll a = n * t / nt, b = n * (t + 1) / nt;
        for (ll q = a; q < b; q++) v[q] = 0.0;
    }
    return v;
}
static double vnorm(const okry* k, const double* w) {
    double s = 0.0;
    const ll n = k->n;
#pragma omp parallel for schedule(static) reduction(+ : s) num_threads(k->nt)
    for (ll q = 0; q < n; q++) s += w[q] * w[q];
    return sqrt(s);
}

/* Right-preconditioned restarted GMRES (CGS2).  fixed_iters > 0: exactly that many iterations (timing samples).
 * hist[0..histcap): residual norms (recurrence; true residual at restarts).  info[0]=iters, [1]=res0, [2]=res, [3]=seconds,
 * [4]=seconds in SpMV, [5]=PC, [6]=orthogonalisation.  Returns the reference's fail flag (DALinearEqn.C:422-434). */
static double g_max_seconds = 0.0;  /* > 0: okry_gmres stops iterating (closing the current cycle) when this wall time is exceeded */
void okry_set_max_seconds(double s) { g_max_seconds = s; }
int okry_gmres(okry* k, const double* rhs, double* x, int restart, int maxit, double rtol, double atol, double tol_diff, int fixed_iters,
               double* hist, int histcap, double* info) {
    const ll n = k->n;
    const int nt = k->nt;
    const double tstart = wall();
    k->t_spmv = k->t_pc = k->t_orth = 0.0; k->c_spmv = k->c_pc = 0;
    const int budget = fixed_iters > 0 ? fixed_iters : maxit;
    int m = restart < budget ? restart : budget;
    if (m < 1) m = 1;
    double** V = (double**)calloc(m + 1, sizeof(double*));
    double* w = vec_alloc(k); double* z = vec_alloc(k); double* r = vec_alloc(k);
    double* H = (double*)calloc((size_t)(m + 1) * m, 8);
    double *cs = (double*)calloc(m, 8), *sn = (double*)calloc(m, 8), *g = (double*)calloc(m + 1, 8), *y = (double*)calloc(m, 8);
    double *h = (double*)calloc(m + 1, 8), *h2 = (double*)calloc(m + 1, 8), *part = (double*)calloc((size_t)nt * (m + 1), 8);
    int nalloc = 0, its = 0, nh = 0, rc = 0, timed_out = 0;
#pragma omp parallel for schedule(static) num_threads(nt)
    for (ll q = 0; q < n; q++) { x[q] = 0.0; r[q] = rhs[q]; }
    double beta = vnorm(k, r);
    const double res0 = beta;
    double res = beta;
    if (hist && nh < histcap) hist[nh++] = beta;
    const double target = fmax(rtol * res0, atol);
    int done = (beta <= target && fixed_iters <= 0) || beta == 0.0;
    while (!done) {
        int mm = budget - its < m ? budget - its : m;
        if (mm <= 0) break;
        if (!V[0]) { V[0] = vec_alloc(k); nalloc = 1; }
#pragma omp parallel for schedule(static) num_threads(nt)
        for (ll q = 0; q < n; q++) V[0][q] = r[q] / beta;
        memset(g, 0, (m + 1) * 8);
        g[0] = beta;
        int j = 0;
        while (j < mm) {
            okry_pc(k, V[j], z);
            okry_spmv(k, z, w);
            const double t0 = wall();
            multi_dot(k, V, j + 1, w, h, part);
            multi_axpy(k, V, j + 1, h, w);
            multi_dot(k, V, j + 1, w, h2, part);
            multi_axpy(k, V, j + 1, h2, w);
            const double hn = vnorm(k, w);
            if (!V[j + 1]) { V[j + 1] = vec_alloc(k); nalloc = j + 2; if (!V[j + 1]) { rc = -1; goto out; } }
            if (hn > 0) {
                double* vn = V[j + 1];
#pragma omp parallel for schedule(static) num_threads(nt)
                for (ll q = 0; q < n; q++) vn[q] = w[q] / hn;
            }
            k->t_orth += wall() - t0;
            double* Hj = H + (size_t)j * (m + 1); /* column j */
            for (int i = 0; i <= j; i++) Hj[i] = h[i] + h2[i];
            Hj[j + 1] = hn;
            for (int i = 0; i < j; i++) {
                const double t = cs[i] * Hj[i] + sn[i] * Hj[i + 1];
                Hj[i + 1] = -sn[i] * Hj[i] + cs[i] * Hj[i + 1];
                Hj[i] = t;
            }
            const double d = hypot(Hj[j], Hj[j + 1]);
            cs[j] = d > 0 ? Hj[j] / d : 1.0; sn[j] = d > 0 ? Hj[j + 1] / d : 0.0;
            Hj[j] = d; Hj[j + 1] = 0.0;
            g[j + 1] = -sn[j] * g[j];
            g[j] = cs[j] * g[j];
            its++; j++;
            res = fabs(g[j]);
            if (hist && nh < histcap) hist[nh++] = res;
            if (fixed_iters <= 0 && (res <= target || its >= maxit)) break;
            if (hn == 0.0) break;
            if (g_max_seconds > 0.0 && wall() - tstart > g_max_seconds) { timed_out = 1; break; }
        }
        for (int i = j - 1; i >= 0; i--) {
            double s = g[i];
            for (int c = i + 1; c < j; c++) s -= H[(size_t)c * (m + 1) + i] * y[c];
            y[i] = s / H[(size_t)i * (m + 1) + i];
        }
        /* x += M^-1 (V y) */
#pragma omp parallel for schedule(static) num_threads(nt)
        for (ll q = 0; q < n; q++) w[q] = 0.0;
        for (int i = 0; i < j; i++) y[i] = -y[i];
        multi_axpy(k, V, j, y, w);
        okry_pc(k, w, z);
#pragma omp parallel for schedule(static) num_threads(nt)
        for (ll q = 0; q < n; q++) x[q] += z[q];
        okry_spmv(k, x, w);
#pragma omp parallel for schedule(static) num_threads(nt)
        for (ll q = 0; q < n; q++) r[q] = rhs[q] - w[q];
        beta = vnorm(k, r);
        res = beta;
        if (hist && nh > 0) hist[nh - 1] = beta;
        done = fixed_iters > 0 ? its >= fixed_iters : (beta <= target || its >= maxit);
        if (beta == 0.0 || timed_out) done = 1;
    }
    rc = (res0 > 0 && (res / res0 / rtol > tol_diff) && (res / atol > tol_diff)) ? 1 : 0;
out:
    if (info) {
        info[0] = its; info[1] = res0; info[2] = res; info[3] = wall() - tstart; info[4] = k->t_spmv; info[5] = k->t_pc; info[6] = k->t_orth;
        info[7] = nh;
    }
    for (int i = 0; i < nalloc; i++) free(V[i]);
    free(V); free(w); free(z); free(r); free(H); free(cs); free(sn); free(g); free(y); free(h); free(h2); free(part);
    return rc;
}
