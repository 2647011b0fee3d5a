// amd.cpp -- approximate-minimum-degree ordering on a quotient graph (host analyse phase).
//
// Implements the published algorithm of Amestoy, Davis & Duff, "An approximate minimum degree
// ordering algorithm", SIAM J. Matrix Anal. Appl. 17(4), 1996: quotient graph with element
// absorption, approximate external degrees  d_i = min(n-k, d_i + |Lp\i|, |A_i\i| + |Lp\i| +
// sum_{e in E_i\p} |L_e\Lp|), mass elimination, indistinguishable-variable merging by hashing,
// aggressive absorption, dense-row deferral.  Written from the paper for this project; the
// reference (Tulip.jl) gets its ordering from CHOLMOD's AMD ([ext], src/KKT/Cholmod/spd.jl:17).
//
// Input: undirected graph, n nodes, adjacency in CSR form without self loops, both directions
// present.  Output: order[k] = node eliminated k-th.
#include "tlpk_host.hpp"

#include <algorithm>
#include <cmath>

namespace tlpk {

namespace {

struct QuotientGraph {
    i32 n;
    std::vector<i64> pe;      // start of the list of node i in iw, or -1 when dead
    std::vector<i32> len;     // total list length (elements first, then variables)
    std::vector<i32> elen;    // >=0: #elements in a live variable's list; -2: live element; -1: dead
    std::vector<i32> nv;      // supervariable weight; <0 while a member of the current Lp; 0 dead
    std::vector<i64> degree;  // approximate external degree (variables) / |Le| weight (elements)
    std::vector<i64> w;       // element marks: 0 = dead element, >= wflg see scan 1
    std::vector<i32> next, last, head, hhead;
    std::vector<i32> mhead, mtail, mnext;   // member chains of supervariables / elements
    std::vector<i32> iw;
    i64 pfree = 0;

    void compact(i32 keep_alive_hint) {
        (void)keep_alive_hint;
        // Gather live objects (principal variables and live elements), sorted by position.
        std::vector<std::pair<i64, i32>> live;
        live.reserve(n);
        for (i32 i = 0; i < n; ++i)
            if (pe[i] >= 0 && len[i] > 0 && (elen[i] == -2 ? w[i] != 0 : (elen[i] >= 0 && nv[i] != 0)))
                live.emplace_back(pe[i], i);
        std::sort(live.begin(), live.end());
        i64 dst = 0;
        for (auto &pr : live) {
            const i32 i = pr.second;
            const i64 src = pr.first;
            if (src != dst) std::copy(iw.begin() + src, iw.begin() + src + len[i], iw.begin() + dst);
            pe[i] = dst;
            dst += len[i];
        }
        pfree = dst;
    }
};

}  // namespace

void amd_order(i32 n, const std::vector<i64> &xadj, const std::vector<i32> &adj, std::vector<i32> &order) {
    order.clear();
    order.reserve(n);
    if (n == 0) return;
    QuotientGraph g;
    g.n = n;
    g.pe.assign(n, -1); g.len.assign(n, 0); g.elen.assign(n, 0); g.nv.assign(n, 1);
    g.degree.assign(n, 0); g.w.assign(n, 1);
    g.next.assign(n, -1); g.last.assign(n, -1); g.head.assign(n + 1, -1); g.hhead.assign(n, -1);
    g.mhead.resize(n); g.mtail.resize(n); g.mnext.assign(n, -1);
    const i64 nnz = xadj[n];
    g.iw.resize((size_t)(nnz + nnz / 4 + 4 * (i64)n + 64));
    std::copy(adj.begin(), adj.begin() + nnz, g.iw.begin());
    g.pfree = nnz;

    auto &pe = g.pe; auto &len = g.len; auto &elen = g.elen; auto &nv = g.nv; auto &degree = g.degree;
    auto &w = g.w; auto &next = g.next; auto &last = g.last; auto &head = g.head; auto &hhead = g.hhead;
    auto &iw = g.iw;

    // dense-node threshold: such nodes are deferred to the end of the ordering
    const i64 dense = std::max<i64>(16, (i64)(10.0 * std::sqrt((double)n)));
    std::vector<i32> deferred;
    i64 nel = 0;

    auto list_insert = [&](i32 i, i64 d) {
        const i32 h = head[d];
        next[i] = h; last[i] = -1;
        if (h != -1) last[h] = i;
        head[d] = i;
    };
    auto list_remove = [&](i32 i) {
        const i64 d = degree[i];
        if (last[i] != -1) next[last[i]] = next[i]; else head[d] = next[i];
        if (next[i] != -1) last[next[i]] = last[i];
    };
    auto absorb_members = [&](i32 into, i32 from) {   // append member chain of `from` to `into`
        g.mnext[g.mtail[into]] = g.mhead[from];
        g.mtail[into] = g.mtail[from];
    };

    for (i32 i = 0; i < n; ++i) {
        g.mhead[i] = g.mtail[i] = i;
        pe[i] = xadj[i];
        len[i] = (i32)(xadj[i + 1] - xadj[i]);
        degree[i] = len[i];
    }
    std::vector<i32> pivots;            // elements in elimination order
    pivots.reserve(n);
    for (i32 i = 0; i < n; ++i) {
        if (degree[i] == 0) {           // isolated: eliminate right away
            elen[i] = -2; pe[i] = -1; w[i] = 0; nel += 1; pivots.push_back(i);
        } else if (degree[i] > dense) { // dense: remove from the graph, order last
            nv[i] = 0; elen[i] = -1; pe[i] = -1; nel += 1; deferred.push_back(i);
        } else {
            list_insert(i, degree[i]);
        }
    }
    i64 wflg = 2;
    i64 mindeg = 0;

    while (nel < n) {
        while (mindeg < n && head[mindeg] == -1) ++mindeg;
        const i32 p = head[mindeg];
        list_remove(p);
        const i32 elenp = elen[p];
        i32 nvpiv = nv[p];
        nel += nvpiv;
        nv[p] = -nvpiv;
        i64 dp = 0;                     // weighted size of Lp
        i64 p_start;
        i32 lp_count = 0;

        if (elenp == 0) {
            // Lp = live variables of A_p, built in place
            p_start = pe[p];
            i64 q = p_start;
            for (i64 k = p_start; k < p_start + len[p]; ++k) {
                const i32 j = iw[k];
                if (nv[j] > 0) {
                    dp += nv[j];
                    nv[j] = -nv[j];
                    iw[q++] = j;
                    list_remove(j);
                }
            }
            lp_count = (i32)(q - p_start);
        } else {
            // need up to degree[p] fresh entries (an upper bound of |Lp|)
            const i64 need = std::min<i64>(degree[p], (i64)n) + 1;
            if (g.pfree + need > (i64)iw.size()) {
                g.compact(p);
                if (g.pfree + need > (i64)iw.size()) iw.resize((size_t)(g.pfree + need + iw.size() / 2));
            }
            p_start = g.pfree;
            i64 q = p_start;
            const i64 pp = pe[p];
            for (i32 t = 0; t <= elenp; ++t) {
                i32 e; i64 s, cnt;
                if (t < elenp) { e = iw[pp + t]; s = pe[e]; cnt = len[e]; if (w[e] == 0 || s < 0) continue; }
                else { e = p; s = pp + elenp; cnt = len[p] - elenp; }
                for (i64 k = s; k < s + cnt; ++k) {
                    const i32 j = iw[k];
                    if (nv[j] > 0) {
                        dp += nv[j];
                        nv[j] = -nv[j];
                        iw[q++] = j;
                        list_remove(j);
                    }
                }
                if (e != p) {           // element e is absorbed into p (its own pivot slot stays)
                    pe[e] = -1; w[e] = 0;
                }
            }
            lp_count = (i32)(q - p_start);
            g.pfree = q;
        }
        degree[p] = dp;
        pe[p] = p_start;
        len[p] = lp_count;
        elen[p] = -2;
        w[p] = 1;                       // live element (any nonzero value < wflg)

        // scan 1: w[e] - wflg = |Le \ Lp| for every element e adjacent to a variable of Lp
        for (i64 k = p_start; k < p_start + lp_count; ++k) {
            const i32 i = iw[k];
            const i32 eln = elen[i];
            if (eln <= 0) continue;
            const i64 nvi = -nv[i];
            const i64 wnvi = wflg - nvi;
            for (i64 q = pe[i]; q < pe[i] + eln; ++q) {
                const i32 e = iw[q];
                if (w[e] >= wflg) w[e] -= nvi;
                else if (w[e] != 0) w[e] = degree[e] + wnvi;
            }
        }
        // scan 2: degree update, list pruning, hashing
        for (i64 k = p_start; k < p_start + lp_count; ++k) {
            const i32 i = iw[k];
            const i64 p1 = pe[i];
            const i64 p2 = p1 + elen[i];
            i64 pn = p1;
            i64 d = 0;
            unsigned long long h = 0;
            for (i64 q = p1; q < p2; ++q) {
                const i32 e = iw[q];
                if (w[e] == 0) continue;            // dead (absorbed) element
                const i64 dext = w[e] - wflg;
                if (dext > 0) { d += dext; iw[pn++] = e; h += (unsigned)e; }
                else {                               // Le subset of Lp: aggressive absorption
                    pe[e] = -1; w[e] = 0;
                }
            }
            const i32 new_elen = (i32)(pn - p1) + 1;
            const i64 p3 = pn;
            for (i64 q = p2; q < p1 + len[i]; ++q) {
                const i32 j = iw[q];
                const i32 nvj = nv[j];
                if (nvj > 0) { d += nvj; iw[pn++] = j; h += (unsigned)j; }
            }
            if (d == 0) {
                // i is adjacent to nothing but Lp: eliminate it together with p
                const i32 nvi = -nv[i];
                dp -= nvi; nvpiv += nvi; nel += nvi;
                nv[i] = 0; elen[i] = -1; pe[i] = -1;
                absorb_members(p, i);
            } else {
                degree[i] = std::min(degree[i], d);
                // make room for p as the first element of the list
                iw[pn] = iw[p3];
                iw[p3] = iw[p1];
                iw[p1] = p;
                len[i] = (i32)(pn - p1) + 1;
                elen[i] = new_elen;
                const i32 hb = (i32)(h % (unsigned long long)n);
                next[i] = hhead[hb];
                hhead[hb] = i;
                last[i] = hb;
            }
        }
        degree[p] = dp;
        wflg += (i64)n + 2;

        // supervariable detection among the members of Lp
        for (i64 k = p_start; k < p_start + lp_count; ++k) {
            const i32 i0 = iw[k];
            if (nv[i0] >= 0) continue;
            const i32 hb = last[i0];
            i32 i = hhead[hb];
            hhead[hb] = -1;
            for (; i != -1 && next[i] != -1; i = next[i], ++wflg) {
                const i32 ln = len[i], eln = elen[i];
                for (i64 q = pe[i] + 1; q < pe[i] + ln; ++q) w[iw[q]] = wflg;
                i32 jlast = i;
                for (i32 j = next[i]; j != -1;) {
                    bool same = (len[j] == ln && elen[j] == eln);
                    for (i64 q = pe[j] + 1; same && q < pe[j] + ln; ++q)
                        if (w[iw[q]] != wflg) same = false;
                    if (same) {
                        nv[i] += nv[j];           // both negative
                        nv[j] = 0; elen[j] = -1; pe[j] = -1;
                        absorb_members(i, j);
                        j = next[j];
                        next[jlast] = j;
                    } else {
                        jlast = j;
                        j = next[j];
                    }
                }
            }
        }
        wflg += 2;
        // finalise Lp: restore weights, final degrees, back into the degree lists
        i64 q = p_start;
        for (i64 k = p_start; k < p_start + lp_count; ++k) {
            const i32 i = iw[k];
            const i32 nvi = -nv[i];
            if (nvi <= 0) continue;
            nv[i] = nvi;
            i64 d = degree[i] + dp - nvi;
            d = std::min<i64>(d, (i64)n - nel - nvi);
            if (d < 0) d = 0;
            degree[i] = d;
            list_insert(i, d);
            if (d < mindeg) mindeg = d;
            iw[q++] = i;
        }
        nv[p] = nvpiv;
        len[p] = (i32)(q - p_start);
        if (len[p] == 0) { pe[p] = -1; w[p] = 0; }
        if (elenp != 0) g.pfree = q;
        pivots.push_back(p);
    }

    // Elimination order: pivots in order, each followed by the variables that were merged into
    // it (indistinguishable variables) or eliminated with it (mass elimination).  Absorbed
    // elements keep their own, earlier pivot slot.
    std::vector<char> seen(n, 0);
    for (i32 p : pivots)
        for (i32 v = g.mhead[p]; v != -1; v = g.mnext[v])
            if (!seen[v]) { seen[v] = 1; order.push_back(v); }
    for (i32 v : deferred) if (!seen[v]) { seen[v] = 1; order.push_back(v); }
    for (i32 v = 0; v < n; ++v) if (!seen[v]) order.push_back(v);   // safety net, never expected
}

}  // namespace tlpk
