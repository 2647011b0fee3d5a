// blocks.cpp -- tlpk_detect_blocks: find the block-angular structure of a constraint matrix.
//
// Why: the block-angular hook of the backend (tlpk_options.row_block) needs one block id per ROW OF THE MATRIX THAT
// KKT.setup RECEIVES.  With Tulip's default Presolve level that matrix is the presolved, rescaled one
// (/root/reference/src/model.jl:88-131, /root/reference/src/Presolve/Presolve.jl:177-305): rows have been removed and
// renumbered, the user never sees it.  An explicit map can only be written for Presolve_Level = 0; the detection below works
// on whatever matrix arrives (Backend(row_block = :auto) in the Julia glue, tlpk_options.detect_blocks = 1 in the C ABI).
//
// What: rows are vertices, two rows are adjacent when they share a column.  Remove the k DENSEST rows ("linking" rows: a
// linking row of a block-angular LP touches columns of many blocks, a block row a handful of its own block's columns) and
// take the connected components of the rest.  The removed sets are nested in k, so the size of the largest component is
// non-increasing in k: a small k <= max_link_rows for which no component holds more than half of the remaining rows (or: at most
// 80 % next to a second component of block size -- LPs with one dominant block) is found by a geometric probe + bisection (each probe is
// one union-find pass over the columns, O(nnz); the remaining-row count m - k shrinks too, so the acceptance test is not strictly
// monotone and the bisection returns a valid k, not necessarily the smallest).  Then
//   * removed rows whose columns all lie in ONE component (or in none) go back into a block (least dense first),
//   * components are packed into blocks: a component with at least 1/64 of the rows of the largest is a block of its own,
//     the small ones (isolated rows: an inequality row that only holds its slack) are dealt to the currently smallest
//     block -- any union of components is a valid diagonal block,
//   * blocks are numbered by their first row, so that contiguous block ranges (the sharding unit) follow the row order.
// No structure (fewer than two blocks): n_blocks = 1, every row in block 0 -- the caller then takes the general sparse path.
#include <algorithm>
#include <cstdlib>
#include <cstdint>
#include <functional>
#include <new>
#include <numeric>
#include <queue>
#include <vector>

#include "../../include/tlpk.h"

namespace {

using i64 = int64_t;
using i32 = int32_t;

struct UnionFind {
    std::vector<i32> p;
    explicit UnionFind(i32 n) : p((size_t)n) { std::iota(p.begin(), p.end(), 0); }
    i32 find(i32 x) { while (p[x] != x) { p[x] = p[p[x]]; x = p[x]; } return x; }
    void unite(i32 a, i32 b) { a = find(a); b = find(b); if (a != b) p[std::max(a, b)] = std::min(a, b); }
};

}  // namespace

extern "C" int tlpk_detect_blocks(int64_t m64, int64_t n64, const int64_t *colptr, const int64_t *rowval, int index_base,
                                  int64_t max_link_rows, int64_t *row_block, int64_t *n_blocks, int64_t *n_link) {
    if (m64 < 0 || n64 < 0 || !colptr || !row_block || (index_base != 0 && index_base != 1)) return TLPK_BADARG;
    if (m64 >= ((i64)1 << 31) || n64 >= ((i64)1 << 31)) return TLPK_TOO_LARGE;
    const i32 m = (i32)m64, n = (i32)n64;
    const i64 nnz = colptr[n] - index_base;
    if (nnz < 0 || (nnz > 0 && !rowval)) return TLPK_BADARG;
    if (n_blocks) *n_blocks = 1;
    if (n_link) *n_link = 0;
    for (i32 i = 0; i < m; ++i) row_block[i] = 0;
    if (m < 2) return TLPK_OK;
    try {
        // row counts, CSR (rows -> columns)
        std::vector<i64> tp((size_t)m + 1, 0);
        for (i32 j = 0; j < n; ++j) {
            if (colptr[j + 1] < colptr[j]) return TLPK_BADARG;
            for (i64 p = colptr[j] - index_base; p < colptr[j + 1] - index_base; ++p) {
                const i64 r = rowval[p] - index_base;
                if (r < 0 || r >= m) return TLPK_BADARG;
                ++tp[(size_t)r + 1];
            }
        }
        std::vector<i32> cnt((size_t)m);
        for (i32 i = 0; i < m; ++i) { cnt[i] = (i32)tp[(size_t)i + 1]; tp[(size_t)i + 1] += tp[(size_t)i]; }
        std::vector<i32> tj((size_t)nnz);
        {
            std::vector<i64> cur(tp.begin(), tp.end() - 1);
            for (i32 j = 0; j < n; ++j)
                for (i64 p = colptr[j] - index_base; p < colptr[j + 1] - index_base; ++p) tj[(size_t)cur[(size_t)(rowval[p] - index_base)]++] = j;
        }
        // rows by decreasing density (ties: by index -- deterministic)
        std::vector<i32> order((size_t)m);
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](i32 a, i32 b) { return cnt[a] > cnt[b]; });
        i64 kmax = max_link_rows > 0 ? max_link_rows : std::max<i64>(64, m / 20);
        kmax = std::min<i64>(kmax, m - 2);
        std::vector<char> link((size_t)m, 0);
        std::vector<i32> comp((size_t)m), csize;
        i64 second = 0;
        // one probe: components of the rows that are not among the k densest; returns the size of the largest
        auto probe = [&](i64 k) -> i64 {
            std::fill(link.begin(), link.end(), 0);
            for (i64 t = 0; t < k; ++t) link[(size_t)order[(size_t)t]] = 1;
            UnionFind uf(m);
            for (i32 j = 0; j < n; ++j) {
                i32 first = -1;
                for (i64 p = colptr[j] - index_base; p < colptr[j + 1] - index_base; ++p) {
                    const i32 r = (i32)(rowval[p] - index_base);
                    if (link[(size_t)r]) continue;
                    if (first < 0) first = r; else uf.unite(first, r);
                }
            }
            csize.assign((size_t)m, 0);
            i64 best = 0;
            for (i32 i = 0; i < m; ++i) {
                if (link[(size_t)i]) { comp[(size_t)i] = -1; continue; }
                const i32 r = uf.find(i);
                comp[(size_t)i] = r;
                best = std::max<i64>(best, ++csize[(size_t)r]);
            }
            second = 0;                             // size of the second largest component
            bool seen_best = false;
            for (i32 i = 0; i < m; ++i) {
                const i64 c = csize[(size_t)i];
                if (c == best && !seen_best) { seen_best = true; continue; }
                second = std::max(second, c);
            }
            return best;
        };
        // A set of k linking rows is accepted when no component holds more than half of the remaining rows, or -- a block-angular LP with one
        // DOMINANT block (two blocks at 60 / 40, one large block and several smaller ones; round-3 advisor finding) -- when the largest holds at most
        // max_fraction of them AND a second component of block size exists (one giant component + isolated rows is not a structure).
        // TLPK_DETECT_MAX_FRACTION (default 0.8) tunes it.
        static const double max_fraction = [] { const char *e = std::getenv("TLPK_DETECT_MAX_FRACTION"); const double v = e ? std::atof(e) : 0.8; return std::min(0.95, std::max(0.5, v)); }();
        auto good = [&](i64 k) {
            const i64 best = probe(k);
            if (2 * best <= (m - k)) return true;
            return (double)best <= max_fraction * (double)(m - k) && second >= std::max<i64>(32, best / 64);
        };
        i64 lo = -1, hi = -1;                       // lo: largest k known bad, hi: smallest k known good
        for (i64 k = 0;; k = (k == 0) ? 1 : std::min(kmax, 4 * k)) {
            if (good(k)) { hi = k; break; }
            lo = k;
            if (k >= kmax) break;
        }
        if (hi < 0) return TLPK_OK;                 // no block structure within the budget of linking rows
        while (hi - lo > 1) { const i64 mid = lo + (hi - lo) / 2; if (good(mid)) hi = mid; else lo = mid; }
        probe(hi);                                  // comp / link / csize of the accepted k
        // removed rows whose columns see at most one component rejoin it (least dense first: the true linking rows stay)
        std::vector<i32> col_comp((size_t)n, -1);   // component of the block rows of a column (-1: only removed rows so far)
        for (i32 j = 0; j < n; ++j)
            for (i64 p = colptr[j] - index_base; p < colptr[j + 1] - index_base; ++p) {
                const i32 r = (i32)(rowval[p] - index_base);
                if (!link[(size_t)r]) { col_comp[(size_t)j] = comp[(size_t)r]; break; }
            }
        i32 smallest = -1;                          // a home for removed rows that touch no block at all
        for (i32 i = 0; i < m; ++i) if (!link[(size_t)i] && (smallest < 0 || csize[(size_t)comp[(size_t)i]] < csize[(size_t)smallest])) smallest = comp[(size_t)i];
        for (i64 t = hi - 1; t >= 0; --t) {
            const i32 r = order[(size_t)t];
            i32 c = -1; bool many = false;
            for (i64 q = tp[(size_t)r]; q < tp[(size_t)r + 1] && !many; ++q) {
                const i32 cc = col_comp[(size_t)tj[(size_t)q]];
                if (cc < 0) continue;
                if (c < 0) c = cc; else if (c != cc) many = true;
            }
            if (many) continue;
            if (c < 0) c = smallest;
            link[(size_t)r] = 0; comp[(size_t)r] = c; ++csize[(size_t)c];
            for (i64 q = tp[(size_t)r]; q < tp[(size_t)r + 1]; ++q) col_comp[(size_t)tj[(size_t)q]] = c;
        }
        // pack the components into blocks
        std::vector<i32> roots;
        for (i32 i = 0; i < m; ++i) if (!link[(size_t)i] && comp[(size_t)i] == i) roots.push_back(i);     // root = smallest row of its component
        i32 largest = 0;
        for (i32 r : roots) largest = std::max(largest, csize[(size_t)r]);
        const i32 big_min = std::max<i32>(32, largest / 64);
        std::vector<i32> block_of_root((size_t)m, -1);
        i32 nb = 0;
        for (i32 r : roots) if (csize[(size_t)r] >= big_min) block_of_root[(size_t)r] = nb++;              // numbered by first row
        if (nb < 2) {
            // many small components and at most one big one: deal everything into up to 64 bins
            nb = (i32)std::min<size_t>(64, roots.size());
            if (nb < 2) return TLPK_OK;
            std::fill(block_of_root.begin(), block_of_root.end(), -1);
        }
        {
            using Item = std::pair<i64, i32>;       // (rows in the block, block id): smallest first, ties by id
            std::priority_queue<Item, std::vector<Item>, std::greater<Item>> heap;
            std::vector<i64> bsz((size_t)nb, 0);
            for (i32 r : roots) if (block_of_root[(size_t)r] >= 0) bsz[(size_t)block_of_root[(size_t)r]] += csize[(size_t)r];
            for (i32 b = 0; b < nb; ++b) heap.push({bsz[(size_t)b], b});
            std::vector<i32> small;
            for (i32 r : roots) if (block_of_root[(size_t)r] < 0) small.push_back(r);
            std::stable_sort(small.begin(), small.end(), [&](i32 a, i32 b) { return csize[(size_t)a] > csize[(size_t)b]; });
            for (i32 r : small) {
                Item it = heap.top(); heap.pop();
                block_of_root[(size_t)r] = it.second;
                heap.push({it.first + csize[(size_t)r], it.second});
            }
        }
        i64 nl = 0;
        for (i32 i = 0; i < m; ++i) {
            if (link[(size_t)i]) { row_block[i] = -1; ++nl; }
            else row_block[i] = block_of_root[(size_t)comp[(size_t)i]];
        }
        if (n_blocks) *n_blocks = nb;
        if (n_link) *n_link = nl;
        return TLPK_OK;
    } catch (const std::bad_alloc &) {
        for (i32 i = 0; i < m; ++i) row_block[i] = 0;
        return TLPK_OOM;
    }
}
