// Jacobian connectivity (dRdWCon), colouring and assembly maps - host graph work, one-off per mesh.
#pragma once
#include <cstdlib>
#include <functional>
#include <new>
#include <sys/mman.h>
#include <memory>
#include <type_traits>
#include <utility>
#include "das_common.hpp"

namespace das {

enum StateKind { KIND_VEC = 0, KIND_SCL = 1, KIND_FACE = 2 };

struct StateDef {
    std::string name;
    StateKind kind;
    long long offset;  // DAIndex "state" ordering offset (reference DAIndex.C:188-258)
    long long size;
};

// stateResConInfo: per residual block, per level, bitmask over states
struct Stencil {
    std::vector<StateDef> states;
    std::vector<std::vector<unsigned>> levels;  // levels[resBlock][lv] = state bitmask
    long long n = 0;
};

Stencil make_stencil(int solver, int nC, int nF, const Options& opt, bool isPC, bool simpleHasT = false);

// std::vector whose resize() leaves trivially-constructible elements uninitialised: the big index arrays are written
// in full by parallel loops, and a serial zero-fill (page faults of GBs of fresh memory) used to cost more than the
// loops themselves
// Large blocks (>= 4 MB) are 2 MB-aligned and marked MADV_HUGEPAGE: with transparent huge pages in "madvise" mode the first
// touch of a multi-GB index array takes 512x fewer page faults - on a many-core host the faults (serialised on the process'
// memory-map lock) are what the parallel graph loops otherwise wait for.
template <class T>
struct default_init_allocator : std::allocator<T> {
    template <class U> struct rebind { using other = default_init_allocator<U>; };
    using std::allocator<T>::allocator;
    static constexpr size_t HUGE = size_t(2) << 20;
    T* allocate(size_t cnt) {
        const size_t bytes = cnt * sizeof(T);
        if (bytes < 2 * HUGE) return std::allocator<T>::allocate(cnt);
        const size_t rounded = (bytes + HUGE - 1) / HUGE * HUGE;
        void* p = std::aligned_alloc(HUGE, rounded);
        if (!p) throw std::bad_alloc();
        (void)madvise(p, rounded, MADV_HUGEPAGE);
        return static_cast<T*>(p);
    }
    void deallocate(T* p, size_t cnt) noexcept {
        if (cnt * sizeof(T) < 2 * HUGE) std::allocator<T>::deallocate(p, cnt);
        else std::free(p);
    }
    template <class U> void construct(U* p) noexcept(std::is_nothrow_default_constructible<U>::value) { ::new (static_cast<void*>(p)) U; }
    template <class U, class... A> void construct(U* p, A&&... a) { ::new (static_cast<void*>(p)) U(std::forward<A>(a)...); }
};
template <class T> using uvector = std::vector<T, default_init_allocator<T>>;

struct JacCon {
    long long n = 0, nnz = 0;
    std::vector<long long> rowptr;  // rows = residuals
    uvector<int> col;               // sorted state indices
    std::vector<int> anchor;        // per row: the cell it is anchored at (cell rows: the cell; face rows: owner)
    // transposed structure (rows = states j, cols = residual i), CSR
    std::vector<long long> t_rowptr;
    uvector<int> t_col;
    // coloured assembly: the columns (states) sorted by colour, cl_ptr[c]..cl_ptr[c+1] = the columns of colour c.  After the
    // residual pass of colour c, transposed row j of such a column IS the list of residuals that depend on it, and for each of
    // them j is the unique column of colour c: vals[t_rowptr[j] + q] = dR_{t_col[..]}/dW_j (setPartDerivMat, reference
    // DAPartDeriv.C:109-208) - no per-entry scatter map, no search.
    std::vector<long long> cl_ptr;
    std::vector<int> cl_cols;
    void build(const Mesh& m, const Stencil& st);
    void build_transpose_and_maps(const std::vector<int>& colors);
    void build_transpose();                                   // t_rowptr / t_col (needs no colours: can run beside the colouring)
    void build_colour_lists(const std::vector<int>& colors);  // cl_ptr / cl_cols
};

// distance-2 (column) colouring of `con`: greedy first-fit over the non-dominated rows.
// Returns number of colours.  Validity rule = reference DAColoring.C:931-1037.
// `centres` (3 per anchor cell, optional) enables the deterministic tile-parallel variant on large meshes.
// `device_fn` (optional): called with the kept rows and the CSC over them; if it returns true it has written `colors`
// (the device first-fit of das_color.hpp), otherwise the host variants run.
using ColorDeviceFn = std::function<bool(long long n, const std::vector<long long>& keep, const std::vector<long long>& cptr, const uvector<int>& crow,
                                         const uvector<int>& cpos, const std::vector<long long>& rowptr, const uvector<int>& col,
                                         std::vector<int>& colors)>;
// `graph_fn` (optional): called right after the dominance pruning with the kept rows only - a device path that derives the
// column -> net incidence itself (das_graph.hpp); if it returns true it has written `colors` and no host CSC is built
using ColorGraphFn = std::function<bool(long long n, const std::vector<long long>& keep, std::vector<int>& colors)>;
int d2_coloring(const JacCon& con, std::vector<int>& colors, const double* centres = nullptr, const ColorDeviceFn& device_fn = nullptr,
                const ColorGraphFn& graph_fn = nullptr);
bool validate_coloring(const JacCon& con, const std::vector<int>& colors);

}  // namespace das
