// Jacobian connectivity + colouring (host).
//
// Restates what the reference builds with PETSc matrices:
//   * stencil tables          reference src/adjoint/DAStateInfo/DAStateInfoSimpleFoam.C:78-128,
//                             DAStateInfoScalarTransportFoam.C:69-75, DASpalartAllmaras.C:364-383
//   * PC level reduction      reference src/adjoint/DASolver/DASolver.C:576-705 (maxResConLv4JacPCMat)
//   * row connectivity        reference src/adjoint/DAJacCon/DAJacCon.C:304-667 (addStateConnections: level k =
//                             k-fold face-neighbour expansion; phi at level k = faces of the level-k cells),
//                             :2039-2600 (setupdRdWCon; face rows take both adjacent cells; boundary-face
//                             rows use level k-1 for the phi test, "levelCheck")
//   * colouring validity      reference src/adjoint/DAColoring/DAColoring.C:931-1037
// The pattern is built on the whole (per-GPU) mesh, so the reference's inter-processor bookkeeping
// (stateBoundaryCon, DAJacCon.C:800-1205) has no counterpart.
#include "das_jaccon.hpp"

#include <omp.h>
#include <cstdlib>

#include <algorithm>
#include <numeric>

namespace das {

Stencil make_stencil(int solver, int nC, int nF, const Options& opt, bool isPC, bool simpleHasT) {
    Stencil st;
    auto add_state = [&](const char* nm, StateKind k) {
        StateDef s;
        s.name = nm;
        s.kind = k;
        s.offset = st.n;
        s.size = k == KIND_VEC ? 3LL * nC : (k == KIND_SCL ? (long long)nC : (long long)nF);
        st.n += s.size;
        st.states.push_back(s);
    };
    auto bits = [&](std::initializer_list<const char*> names) {
        unsigned m = 0;
        for (auto nm : names)
            for (size_t i = 0; i < st.states.size(); i++)
                if (st.states[i].name == nm) m |= 1u << i;
        return m;
    };
    if (solver == DAS_SOLVER_SIMPLEFOAM && simpleHasT) {
        // DASimpleFoam with the optional T field (DAStateInfoSimpleFoam.C:118-131).  TRes lists U at level 0 in addition
        // to the reference's table: alphat_b = nut_b/Prt of a wall-function face depends on the cell velocity.
        add_state("U", KIND_VEC);
        add_state("p", KIND_SCL);
        add_state("T", KIND_SCL);
        add_state("nuTilda", KIND_SCL);
        add_state("phi", KIND_FACE);
        st.levels.resize(5);
        st.levels[0] = {bits({"U", "p", "nuTilda", "phi"}), bits({"U", "p", "nuTilda"}), bits({"U"})};  // URes
        st.levels[1] = {bits({"U", "p", "nuTilda", "phi"}), bits({"U", "p", "nuTilda", "phi"}), bits({"U", "p", "nuTilda"}),
                        bits({"U"})};                                                                // pRes
        st.levels[2] = {bits({"U", "T", "nuTilda", "phi"}), bits({"T", "nuTilda"}), bits({"T"})};      // TRes
        st.levels[3] = {bits({"U", "nuTilda", "phi"}), bits({"U", "nuTilda"}), bits({"nuTilda"})};    // nuTildaRes
        st.levels[4] = {bits({"U", "p", "nuTilda", "phi"}), bits({"U", "p", "nuTilda"}), bits({"U"})};  // phiRes
    } else if (solver == DAS_SOLVER_SIMPLEFOAM) {
        // order: volVector, volScalar, model, surfaceScalar (reference DAIndex.C:43-63)
        add_state("U", KIND_VEC);
        add_state("p", KIND_SCL);
        add_state("nuTilda", KIND_SCL);
        add_state("phi", KIND_FACE);
        st.levels.resize(4);
        st.levels[0] = {bits({"U", "p", "nuTilda", "phi"}), bits({"U", "p", "nuTilda"}), bits({"U"})};  // URes
        st.levels[1] = {bits({"U", "p", "nuTilda", "phi"}), bits({"U", "p", "nuTilda", "phi"}), bits({"U", "p", "nuTilda"}),
                        bits({"U"})};                                                                // pRes
        st.levels[2] = {bits({"U", "nuTilda", "phi"}), bits({"U", "nuTilda"}), bits({"nuTilda"})};    // nuTildaRes
        st.levels[3] = {bits({"U", "p", "nuTilda", "phi"}), bits({"U", "p", "nuTilda"}), bits({"U"})};  // phiRes
    } else if (DAS_IS_COMPRESSIBLE(solver)) {  // DAStateInfoTurboFoam.C:82-119 carries the same level table
        // DAStateInfoRhoSimpleFoam.C:40-47,79-116 and the compressible SA table DASpalartAllmaras.C:364-383
        add_state("U", KIND_VEC);
        add_state("p", KIND_SCL);
        add_state("T", KIND_SCL);
        add_state("nuTilda", KIND_SCL);
        add_state("phi", KIND_FACE);
        st.levels.resize(5);
        st.levels[0] = {bits({"U", "p", "T", "nuTilda", "phi"}), bits({"U", "p", "T", "nuTilda"}), bits({"U", "T"})};                        // URes
        st.levels[1] = {bits({"U", "p", "T", "nuTilda", "phi"}), bits({"U", "p", "T", "nuTilda", "phi"}), bits({"U", "p", "T", "nuTilda"}),
                        bits({"U"})};                                                                                                   // pRes
        st.levels[2] = {bits({"U", "p", "T", "nuTilda", "phi"}), bits({"U", "p", "T", "nuTilda"}), bits({"U", "p", "T"})};                  // TRes
        st.levels[3] = {bits({"U", "T", "p", "nuTilda", "phi"}), bits({"U", "T", "p", "nuTilda"}), bits({"T", "p", "nuTilda"})};            // nuTildaRes
        st.levels[4] = {bits({"U", "p", "T", "nuTilda", "phi"}), bits({"U", "p", "T", "nuTilda"}), bits({"U", "T"})};                       // phiRes
    } else if (solver == DAS_SOLVER_SCALARTRANSPORTFOAM) {
        add_state("T", KIND_SCL);
        st.levels.resize(1);
        st.levels[0] = {bits({"T"}), bits({"T"}), bits({"T"})};
    } else {
        throw Error(DAS_ERR_ARG, "unknown solver id");
    }
    if (isPC) {
        for (size_t b = 0; b < st.states.size(); b++) {
            std::string key = "maxResConLv4JacPCMat." + st.states[b].name + "Res";
            long long mx = opt.geti(key);  // throws if absent, like DASolver.C:635-664
            DAS_CHECK(mx >= 0 && mx < (long long)st.levels[b].size(), DAS_ERR_ARG,
                      "maxResConLv4JacPCMat level larger than stateResConInfo level for " + key);
            st.levels[b].resize(mx + 1);
        }
    }
    return st;
}

namespace {
struct RowBuilder {
    const Mesh& m;
    const Stencil& st;
    std::vector<int> cstamp, fstamp;
    std::vector<unsigned char> cmask;
    std::vector<int> cells, faces;
    std::vector<int> L[4];
    std::vector<int> lstamp;
    int tag = 0, ltag = 0;
    unsigned faceBits = 0;
    RowBuilder(const Mesh& m_, const Stencil& st_) : m(m_), st(st_) {
        cstamp.assign(m.nC, -1);
        fstamp.assign(m.nF, -1);
        cmask.assign(m.nC, 0);
        lstamp.assign(m.nC, -1);
        for (size_t i = 0; i < st.states.size(); i++)
            if (st.states[i].kind == KIND_FACE) faceBits |= 1u << i;
    }
    void rings(int c, int maxlv) {
        L[0].assign(1, c);
        for (int k = 1; k <= maxlv; k++) {
            L[k].clear();
            ltag++;
            for (int x : L[k - 1])
                for (int s = m.cc_ptr[x]; s < m.cc_ptr[x + 1]; s++) {
                    int y = m.cc[s];
                    if (lstamp[y] != ltag) { lstamp[y] = ltag; L[k].push_back(y); }
                }
        }
    }
    void begin() { tag++; cells.clear(); faces.clear(); }
    void touch_cell(int x, unsigned bitsCell) {
        if (cstamp[x] != tag) { cstamp[x] = tag; cmask[x] = 0; cells.push_back(x); }
        cmask[x] |= (unsigned char)bitsCell;
    }
    void touch_faces_of(int x) {
        for (int s = m.cf_ptr[x]; s < m.cf_ptr[x + 1]; s++) {
            int f = m.cf_face[s] & 0x7fffffff;
            if (fstamp[f] != tag) { fstamp[f] = tag; faces.push_back(f); }
        }
    }
    // add the level tables of residual block rb anchored at cell c; bfaceRow: boundary-face levelCheck
    void add(int rb, int c, bool bfaceRow) {
        const auto& lv = st.levels[rb];
        int maxlv = (int)lv.size() - 1;
        rings(c, maxlv);
        for (int k = 0; k <= maxlv; k++) {
            unsigned cb = lv[k] & ~faceBits;
            unsigned fb = (bfaceRow && k > 0) ? (lv[k - 1] & faceBits) : (lv[k] & faceBits);
            for (int x : L[k]) {
                if (cb) touch_cell(x, cb);
                if (fb) touch_faces_of(x);
            }
        }
    }
    void emit(std::vector<int>& out) {
        std::sort(cells.begin(), cells.end());
        std::sort(faces.begin(), faces.end());
        for (size_t i = 0; i < st.states.size(); i++) {
            const StateDef& s = st.states[i];
            if (s.kind == KIND_FACE) {
                if (!faces.empty())  // one face state only (phi)
                    for (int f : faces) out.push_back((int)(s.offset + f));
            } else if (s.kind == KIND_VEC) {
                for (int x : cells)
                    if (cmask[x] & (1u << i)) { out.push_back((int)(s.offset + 3LL * x)); out.push_back((int)(s.offset + 3LL * x + 1)); out.push_back((int)(s.offset + 3LL * x + 2)); }
            } else {
                for (int x : cells)
                    if (cmask[x] & (1u << i)) out.push_back((int)(s.offset + x));
            }
        }
    }
};
}  // namespace

void JacCon::build(const Mesh& m, const Stencil& st) {
    n = st.n;
    DAS_CHECK(n < 2147483647LL, DAS_ERR_ARG, "state count exceeds int32 column indices");
    rowptr.assign(n + 1, 0);
    anchor.assign(n, 0);
    // work items: (state block, entity index) in row order; rows of one item are consecutive
    struct Item { int block; int ent; long long row0; };
    // chunks of entities per block, processed in parallel, concatenated in order
    const int nthreads = std::max(1, das::host_threads());
    struct Chunk { int block; int e0, e1; long long row0; uvector<int> col; std::vector<int> rowlen; };
    std::vector<Chunk> chunks;
    {
        long long r = 0;
        for (size_t b = 0; b < st.states.size(); b++) {
            const StateDef& s = st.states[b];
            const int nent = s.kind == KIND_FACE ? m.nF : m.nC;
            const int ncomp = s.kind == KIND_VEC ? 3 : 1;
            const int nch = std::max(1, std::min(nent, nthreads * 4));
            for (int c = 0; c < nch; c++) {
                Chunk ch;
                ch.block = (int)b;
                ch.e0 = (int)((long long)nent * c / nch);
                ch.e1 = (int)((long long)nent * (c + 1) / nch);
                ch.row0 = r + (long long)ch.e0 * ncomp;
                chunks.push_back(std::move(ch));
            }
            r += (long long)nent * ncomp;
        }
        DAS_CHECK(r == n, DAS_ERR_INTERNAL, "row count mismatch in JacCon::build");
    }
#pragma omp parallel num_threads(das::host_threads())
    {
        RowBuilder rb(m, st);
        std::vector<int> row;
#pragma omp for schedule(dynamic, 1)
        for (long long ci = 0; ci < (long long)chunks.size(); ci++) {
            Chunk& ch = chunks[ci];
            const StateDef& s = st.states[ch.block];
            const int ncomp = s.kind == KIND_VEC ? 3 : 1;
            ch.col.reserve((size_t)(ch.e1 - ch.e0) * ncomp * 160);
            for (int e = ch.e0; e < ch.e1; e++) {
                rb.begin();
                int anch;
                if (s.kind == KIND_FACE) {
                    // a cyclic boundary face is coupled to two cells like an internal face
                    const int cyc = e >= m.nIF ? m.cyc_face[e - m.nIF] : -1;
                    bool bnd = e >= m.nIF && cyc < 0;
                    int cn = bnd ? m.owner[e] : (cyc >= 0 ? m.owner[cyc] : m.neighbour[e]);
                    rb.add(ch.block, cn, bnd);
                    if (!bnd) rb.add(ch.block, m.owner[e], false);
                    anch = m.owner[e];
                } else {
                    rb.add(ch.block, e, false);
                    anch = e;
                }
                row.clear();
                rb.emit(row);
                for (int k = 0; k < ncomp; k++) {
                    ch.col.insert(ch.col.end(), row.begin(), row.end());
                    ch.rowlen.push_back((int)row.size());
                    anchor[ch.row0 + (long long)(e - ch.e0) * ncomp + k] = anch;
                }
            }
        }
    }
    // concatenate
    long long total = 0;
    for (auto& ch : chunks) total += (long long)ch.col.size();
    col.resize(total);
    {
        long long off = 0;
        std::vector<long long> choff(chunks.size());
        for (size_t ci = 0; ci < chunks.size(); ci++) { choff[ci] = off; off += (long long)chunks[ci].col.size(); }
#pragma omp parallel for schedule(dynamic, 1) num_threads(das::host_threads())
        for (long long ci = 0; ci < (long long)chunks.size(); ci++) {
            Chunk& ch = chunks[ci];
            std::copy(ch.col.begin(), ch.col.end(), col.begin() + choff[ci]);
            long long o = choff[ci];
            for (size_t k = 0; k < ch.rowlen.size(); k++) { rowptr[ch.row0 + (long long)k] = o; o += ch.rowlen[k]; }
            uvector<int>().swap(ch.col);
        }
    }
    rowptr[n] = total;
    nnz = total;
}

static bool is_subset(const int* a, long long na, const int* b, long long nb) {
    long long i = 0, j = 0;
    while (i < na && j < nb) {
        if (a[i] == b[j]) { i++; j++; }
        else if (a[i] > b[j]) j++;
        else return false;
    }
    return i == na;
}

int d2_coloring(const JacCon& con, std::vector<int>& colors, const double* centres, const ColorDeviceFn& device_fn, const ColorGraphFn& graph_fn) {
    const long long n = con.n;
    colors.assign(n, -1);
    const bool dbgT = getenv("DAS_DEBUG_TIMING") != nullptr;
    double tq = wall_seconds();
    auto lap = [&](const char* what) { if (dbgT) { double t2 = wall_seconds(); fprintf(stderr, "[dafoam_amd]   colouring: %s %.2f s\n", what, t2 - tq); tq = t2; } };
    // dominance pruning: a row that is a subset of the longest row anchored at the same cell adds no
    // colouring constraint (for the reference's tables every row of a cell / owned face is a subset
    // of that cell's pRes row).  Generic: the subset test decides, no solver-specific assumption.
    long long nAnch = 0;
#pragma omp parallel for reduction(max : nAnch) schedule(static) num_threads(das::host_threads())
    for (long long r = 0; r < n; r++) nAnch = std::max<long long>(nAnch, con.anchor[r] + 1);
    std::vector<long long> dom(nAnch, -1);
    for (long long r = 0; r < n; r++) {  // (serial: rows of one anchor are not contiguous across the state blocks)
        long long len = con.rowptr[r + 1] - con.rowptr[r];
        long long& d = dom[con.anchor[r]];
        if (d < 0 || len > con.rowptr[d + 1] - con.rowptr[d]) d = r;
    }
    std::vector<long long> keep;
    {
        // the subset tests are independent per row: flags in parallel, then the kept rows in ascending order
        std::vector<unsigned char> kept(n, 0);
#pragma omp parallel for schedule(dynamic, 4096) num_threads(das::host_threads())
        for (long long r = 0; r < n; r++) {
            long long len = con.rowptr[r + 1] - con.rowptr[r];
            if (!len) continue;
            long long d = dom[con.anchor[r]];
            kept[r] = !(d != r && is_subset(&con.col[con.rowptr[r]], len, &con.col[con.rowptr[d]], con.rowptr[d + 1] - con.rowptr[d]));
        }
        keep.reserve(nAnch * 2);
        for (long long r = 0; r < n; r++) if (kept[r]) keep.push_back(r);
    }
    lap("prune");
    if (graph_fn && graph_fn(n, keep, colors)) {
        lap("device graph + first-fit");
        int ncd = 0;
        for (long long j = 0; j < n; j++) ncd = std::max(ncd, colors[j] + 1);
        return ncd;
    }
    // CSC over kept rows: stable chunked counting (like JacCon::build_transpose_and_maps), int row ids, no zero-fill
    std::vector<long long> cptr(n + 1, 0);
    uvector<int> crow, cpos;  // cpos: position of the column inside the (ascending) column list of that kept row
    {
        const long long nk = (long long)keep.size();
        const int T = (int)std::max<long long>(1, std::min<long long>({(long long)das::host_threads(), 32LL, nk}));
        std::vector<long long> q0(T + 1);
        for (int t = 0; t <= T; t++) q0[t] = nk * t / T;
        std::vector<uvector<int>> cnt(T);
#pragma omp parallel for schedule(static, 1) num_threads(T)
        for (int t = 0; t < T; t++) {
            cnt[t].resize(n);
            std::fill(cnt[t].begin(), cnt[t].end(), 0);
            for (long long q = q0[t]; q < q0[t + 1]; q++)
                for (long long k = con.rowptr[keep[q]]; k < con.rowptr[keep[q] + 1]; k++) cnt[t][con.col[k]]++;
        }
        // per column: total over the chunks (-> cptr by a serial prefix over n values) and the offset of every chunk inside it
#pragma omp parallel for schedule(static) num_threads(das::host_threads())
        for (long long j = 0; j < n; j++) {
            int acc = 0;
            for (int t = 0; t < T; t++) { int c = cnt[t][j]; cnt[t][j] = acc; acc += c; }
            cptr[j + 1] = acc;
        }
        for (long long j = 0; j < n; j++) cptr[j + 1] += cptr[j];
        crow.resize(cptr[n]);
        if (device_fn) cpos.resize(cptr[n]);
        const bool wantPos = (bool)device_fn;
#pragma omp parallel for schedule(static, 1) num_threads(T)
        for (int t = 0; t < T; t++) {
            uvector<int>& pos = cnt[t];
            for (long long q = q0[t]; q < q0[t + 1]; q++) {
                const long long r = keep[q];
                for (long long k = con.rowptr[r]; k < con.rowptr[r + 1]; k++) {
                    const int j = con.col[k];
                    const long long dst = cptr[j] + pos[j]++;
                    crow[dst] = (int)r;
                    if (wantPos) cpos[dst] = (int)(k - con.rowptr[r]);
                }
            }
        }
    }
    // first-fit colour of column j: the smallest colour not used by any column sharing a kept row with j
    auto color_column = [&](long long j, std::vector<long long>& forb, long long stamp) {
        for (long long q = cptr[j]; q < cptr[j + 1]; q++) {
            const long long r = crow[q];
            for (long long k = con.rowptr[r]; k < con.rowptr[r + 1]; k++) {
                int c = colors[con.col[k]];
                if (c >= 0) {
                    if ((size_t)c >= forb.size()) forb.resize(2 * c + 2, -1);
                    forb[c] = stamp;
                }
            }
        }
        int c = 0;
        while ((size_t)c < forb.size() && forb[c] == stamp) c++;
        colors[j] = c;
    };
    // first-fit over an ascending column list.  Consecutive columns with the same kept-row list (the xyz components of a
    // cell's U) see the same neighbourhood: its forbidden set is gathered once and extended by the colours just handed
    // out - identical to colouring them one by one, at a third of the gathers.
    auto sweep_grouped = [&](const std::vector<long long>& cols, std::vector<long long>& forb) {
        size_t a = 0;
        while (a < cols.size()) {
            const long long j = cols[a];
            color_column(j, forb, j);
            const long long len = cptr[j + 1] - cptr[j];
            size_t b = a + 1;
            while (b < cols.size() && cols[b] == cols[b - 1] + 1 && cptr[cols[b] + 1] - cptr[cols[b]] == len
                   && std::equal(crow.begin() + cptr[j], crow.begin() + cptr[j + 1], crow.begin() + cptr[cols[b]])) {
                const int cprev = colors[cols[b - 1]];
                if ((size_t)cprev >= forb.size()) forb.resize(2 * cprev + 2, -1);
                forb[cprev] = j;  // stamp of the group
                int c = 0;
                while ((size_t)c < forb.size() && forb[c] == j) c++;
                colors[cols[b]] = c;
                b++;
            }
            a = b;
        }
    };
    lap("csc");
    if (device_fn && device_fn(n, keep, cptr, crow, cpos, con.rowptr, con.col, colors)) {
        lap("device first-fit");
        int ncd = 0;
        for (long long j = 0; j < n; j++) ncd = std::max(ncd, colors[j] + 1);
        return ncd;
    }
    const int nth = std::max(1, das::host_threads());
    // Columns are grouped by the cell they live in (anchor) and the cell range is cut into spatially contiguous
    // chunks, so concurrent threads only interact near chunk boundaries (few conflicts, near-serial colour count).
    // DAS_PARALLEL_COLORING=1 selects this speculative variant: ~3.5x faster than serial for ~11 % more colours (451-457
    // vs 410 at 200k cells), but the colour labels depend on thread timing (the Jacobians do not: columns of one colour
    // never share a row).  =0 forces the serial first-fit.  Default: serial below 20k cells, the deterministic
    // tile-parallel variant (further down) above.
    const char* pc_env = getenv("DAS_PARALLEL_COLORING");
    const bool par = nth > 1 && pc_env && pc_env[0] == '1';
    if (par) {
        const int T = std::min(nth, 32);
        std::vector<std::vector<long long>> chunkCols(T);
        for (long long j = 0; j < n; j++) chunkCols[(long long)con.anchor[j] * T / nAnch].push_back(j);
#pragma omp parallel num_threads(T)
        {
            std::vector<long long> forb(4096, -1);
#pragma omp for schedule(static, 1)
            for (int t = 0; t < T; t++)
                for (long long j : chunkCols[t]) color_column(j, forb, j);
        }
        for (int pass = 0; pass < 100; pass++) {
            std::vector<long long> redo;
#pragma omp parallel num_threads(das::host_threads())
            {
                std::vector<long long> seen;  // colour -> column seen in this row
                std::vector<long long> mine;
#pragma omp for schedule(dynamic, 256)
                for (long long q = 0; q < (long long)keep.size(); q++) {
                    long long r = keep[q];
                    seen.assign(seen.size(), -1);
                    for (long long k = con.rowptr[r]; k < con.rowptr[r + 1]; k++) {
                        int j = con.col[k];
                        int c = colors[j];
                        if (c < 0) continue;
                        if ((size_t)c >= seen.size()) seen.resize(2 * c + 2, -1);
                        if (seen[c] >= 0 && seen[c] != j) mine.push_back(std::max<long long>(seen[c], j));
                        else seen[c] = j;
                    }
                }
#pragma omp critical
                redo.insert(redo.end(), mine.begin(), mine.end());
            }
            if (redo.empty()) break;
            std::sort(redo.begin(), redo.end());
            redo.erase(std::unique(redo.begin(), redo.end()), redo.end());
            for (long long j : redo) colors[j] = -1;
            std::vector<long long> forb(4096, -1);
            for (long long j : redo) color_column(j, forb, j);
        }
    } else if (centres && nAnch >= 20000 && nth > 1 && !(pc_env && pc_env[0] == '0')) {
        // Deterministic tile-parallel first-fit.  The anchor cells are cut into compact tiles (recursive coordinate
        // bisection); two tiles are adjacent if some kept row holds columns of both, i.e. if their columns can conflict;
        // the tile graph is coloured greedily into phases and the tiles of one phase are first-fit coloured
        // concurrently.  A column only ever looks at columns of its own or of adjacent tiles, and adjacent tiles are
        // never in the same phase - so there is no race and the result does not depend on the number of threads.
        const long long tileCells = std::max<long long>(2048, (nAnch + 4095) / 4096);
        std::vector<int> tileOf(nAnch, 0);
        int nT = 0;
        {
            std::vector<int> idx(nAnch);
            std::iota(idx.begin(), idx.end(), 0);
            struct Rg { long long b, e; };
            std::vector<Rg> stack{{0, nAnch}};
            while (!stack.empty()) {
                Rg r = stack.back();
                stack.pop_back();
                if (r.e - r.b <= tileCells) {
                    for (long long q = r.b; q < r.e; q++) tileOf[idx[q]] = nT;
                    nT++;
                    continue;
                }
                double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
                for (long long q = r.b; q < r.e; q++)
                    for (int d = 0; d < 3; d++) { double x = centres[3LL * idx[q] + d]; lo[d] = std::min(lo[d], x); hi[d] = std::max(hi[d], x); }
                int ax = 0;
                for (int d = 1; d < 3; d++) if (hi[d] - lo[d] > hi[ax] - lo[ax]) ax = d;
                const long long mid = (r.b + r.e) / 2;
                std::nth_element(idx.begin() + r.b, idx.begin() + mid, idx.begin() + r.e, [&](int a, int b2) {
                    const double xa = centres[3LL * a + ax], xb = centres[3LL * b2 + ax];
                    return xa < xb || (xa == xb && a < b2);
                });
                stack.push_back({mid, r.e});
                stack.push_back({r.b, mid});
            }
        }
        // tile adjacency from the kept rows (bit matrix, OR-merged over threads)
        const size_t words = ((size_t)nT * nT + 63) / 64;
        std::vector<unsigned long long> adjBits(words, 0ULL);
#pragma omp parallel num_threads(das::host_threads())
        {
            std::vector<unsigned long long> mine(words, 0ULL);
            std::vector<int> tiles;
#pragma omp for schedule(dynamic, 1024)
            for (long long q = 0; q < (long long)keep.size(); q++) {
                const long long r = keep[q];
                tiles.clear();
                for (long long k = con.rowptr[r]; k < con.rowptr[r + 1]; k++) {
                    const int t = tileOf[con.anchor[con.col[k]]];
                    if (std::find(tiles.begin(), tiles.end(), t) == tiles.end()) tiles.push_back(t);
                }
                for (size_t a = 0; a < tiles.size(); a++)
                    for (size_t b2 = 0; b2 < tiles.size(); b2++) {
                        const size_t bit = (size_t)tiles[a] * nT + tiles[b2];
                        mine[bit >> 6] |= 1ULL << (bit & 63);
                    }
            }
#pragma omp critical
            for (size_t w = 0; w < words; w++) adjBits[w] |= mine[w];
        }
        auto adjacent = [&](int a, int b2) { const size_t bit = (size_t)a * nT + b2; return (adjBits[bit >> 6] >> (bit & 63)) & 1ULL; };
        std::vector<int> phase(nT, -1);
        int nPh = 0;
        {
            std::vector<int> used;
            for (int t = 0; t < nT; t++) {
                used.assign(nPh + 1, 0);
                for (int u = 0; u < t; u++) if (adjacent(t, u)) used[phase[u]] = 1;
                int p = 0;
                while (used[p]) p++;
                phase[t] = p;
                nPh = std::max(nPh, p + 1);
            }
        }
        std::vector<std::vector<long long>> tileCols(nT);
        for (long long j = 0; j < n; j++) tileCols[tileOf[con.anchor[j]]].push_back(j);
        std::vector<std::vector<int>> phaseTiles(nPh);
        for (int t = 0; t < nT; t++) phaseTiles[phase[t]].push_back(t);
        for (int ph = 0; ph < nPh; ph++) {
            const std::vector<int>& tl = phaseTiles[ph];
#pragma omp parallel num_threads(das::host_threads())
            {
                std::vector<long long> forb(4096, -1);
#pragma omp for schedule(dynamic, 1)
                for (long long ti = 0; ti < (long long)tl.size(); ti++)
                    sweep_grouped(tileCols[tl[ti]], forb);
            }
        }
        if (getenv("DAS_DEBUG_TIMING")) {
            fprintf(stderr, "[dafoam_amd]   colouring: %d tiles of <= %lld cells, %d phases, tiles per phase:", nT, tileCells, nPh);
            for (int ph = 0; ph < nPh; ph++) fprintf(stderr, " %d", (int)phaseTiles[ph].size());
            fprintf(stderr, "\n");
        }
    } else {
        // serial first-fit.  Consecutive columns with the same kept-row list (the xyz components of a cell's U) see the
        // same neighbourhood: its forbidden set is gathered once and extended by the colours just handed out - the
        // result is identical to colouring them one by one, at a third of the gathers.
        std::vector<long long> forb(4096, -1);
        std::vector<long long> all(n);
        std::iota(all.begin(), all.end(), 0LL);
        sweep_grouped(all, forb);
    }
    lap("greedy");
    int ncol = 0;
    for (long long j = 0; j < n; j++) ncol = std::max(ncol, colors[j] + 1);
    DAS_CHECK(ncol < 65535, DAS_ERR_INTERNAL, "more than 65534 colours");
    return ncol;
}

bool validate_coloring(const JacCon& con, const std::vector<int>& colors) {
    // reference DAColoring::validateColoring (DAColoring.C:931-1037): no row holds two columns of one colour.  Rows are independent:
    // every thread checks a contiguous range with its own "colour last seen in row" table (was serial: 1.6 s at 2 M cells)
    int ncol = 0;
    for (long long j = 0; j < con.n; j++) {
        if (colors[j] < 0) return false;
        ncol = std::max(ncol, colors[j] + 1);
    }
    bool ok = true;
#pragma omp parallel num_threads(das::host_threads())
    {
        std::vector<long long> seen((size_t)ncol, -1);
#pragma omp for schedule(static)
        for (long long r = 0; r < con.n; r++) {
            if (!ok) continue;
            for (long long k = con.rowptr[r]; k < con.rowptr[r + 1]; k++) {
                const int c = colors[con.col[k]];
                if (seen[c] == r) {
#pragma omp atomic write
                    ok = false;
                    break;
                }
                seen[c] = r;
            }
        }
    }
    return ok;
}

void JacCon::build_transpose_and_maps(const std::vector<int>& colors) {
    build_transpose();
    build_colour_lists(colors);
}

void JacCon::build_transpose() {
    // Stable parallel counting transpose: the rows are cut into T contiguous chunks; pass 1 counts the entries of every
    // column per chunk, a prefix over (column, chunk) gives each chunk its start inside every transposed row, pass 2
    // lets every chunk fill its rows in ascending order.  Transposed rows come out sorted by residual index and the
    // destination of every pattern entry is known at fill time (no atomics, no per-row sort, no binary search).
    const bool dbg = getenv("DAS_DEBUG_TIMING") != nullptr;
    double tt = wall_seconds();
    auto lap = [&](const char* what) { if (dbg) { double t2 = wall_seconds(); fprintf(stderr, "[dafoam_amd]   maps: %s %.2f s\n", what, t2 - tt); tt = t2; } };
    const int T = (int)std::max<long long>(1, std::min<long long>({(long long)das::host_threads(), 32LL, n}));
    std::vector<long long> r0(T + 1);
    for (int t = 0; t <= T; t++) r0[t] = n * t / T;
    std::vector<uvector<int>> cnt(T);
#pragma omp parallel for schedule(static, 1) num_threads(T)
    for (int t = 0; t < T; t++) {
        cnt[t].resize(n);
        std::fill(cnt[t].begin(), cnt[t].end(), 0);
        for (long long k = rowptr[r0[t]]; k < rowptr[r0[t + 1]]; k++) cnt[t][col[k]]++;
    }
    lap("count");
    t_rowptr.assign(n + 1, 0);
    // cnt[t][j] := offset of chunk t inside transposed row j; row lengths -> t_rowptr by a serial prefix over n values
#pragma omp parallel for schedule(static) num_threads(das::host_threads())
    for (long long j = 0; j < n; j++) {
        int acc = 0;
        for (int t = 0; t < T; t++) { int c = cnt[t][j]; cnt[t][j] = acc; acc += c; }
        t_rowptr[j + 1] = acc;
    }
    for (long long j = 0; j < n; j++) t_rowptr[j + 1] += t_rowptr[j];
    lap("prefix");
    t_col.resize(nnz);
    lap("alloc");
#pragma omp parallel for schedule(static, 1) num_threads(T)
    for (int t = 0; t < T; t++) {
        uvector<int>& pos = cnt[t];
        for (long long r = r0[t]; r < r0[t + 1]; r++)
            for (long long k = rowptr[r]; k < rowptr[r + 1]; k++) {
                const int j = col[k];
                t_col[t_rowptr[j] + pos[j]++] = (int)r;
            }
    }
    cnt.clear();
    lap("fill");
}

void JacCon::build_colour_lists(const std::vector<int>& colors) {
    // the columns sorted by colour (stable counting sort)
    int ncol = 0;
    for (long long jj = 0; jj < n; jj++) ncol = std::max(ncol, colors[jj] + 1);
    cl_ptr.assign(ncol + 1, 0);
    for (long long jj = 0; jj < n; jj++) cl_ptr[colors[jj] + 1]++;
    for (int c = 0; c < ncol; c++) cl_ptr[c + 1] += cl_ptr[c];
    cl_cols.resize(n);
    {
        std::vector<long long> pos(cl_ptr.begin(), cl_ptr.end() - 1);
        for (long long jj = 0; jj < n; jj++) cl_cols[pos[colors[jj]]++] = (int)jj;
    }
}

}  // namespace das
