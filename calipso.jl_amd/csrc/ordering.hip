// ordering.hip — the ordering / symbolic service of the LinearSolver seam: integer work on the host, as in the reference.
//
// The reference factors  P K P' = L D L'  with P = amd(K) from the third-party AMD.jl (qdldl.jl:135), builds the permuted upper-triangular
// matrix and the entry map AtoPAPt (permute_symmetric, qdldl.jl:642-742), the elimination tree and the column counts of L
// (QDLDL_etree!, qdldl.jl:358-395).  Here:
//   calipso_hip_ordering         an elimination order for a sparse symmetric pattern: reverse Cuthill-McKee (minimises the bandwidth — what the
//                                device LDL^T of ldl.hip exploits: everything outside the band is skipped) or minimum degree on the quotient
//                                graph (the fill-reducing order class of AMD; AMD.jl's exact permutation is not reproducible and no reference
//                                test pins it).
//   calipso_hip_symbolic         P A P' (upper triangle, CSC, entries in the reference's placement order), AtoPAPt, etree, column counts — integer
//                                results, bit-exact against the oracle's restatement for the same permutation (tests/test_oracle_qdldl.py,
//                                tests/test_abi_cpu.py::test_symbolic_matches_the_oracle).  Pure host functions: no device needed.
//   calipso_hip_ldl_analyze_csc  chooses / installs the order of a calipso_hip_ldl_* handle; the following factorisations scatter P K P' and, when the
//                                permuted matrix is banded, run the band-limited device factorisation and triangular solves.
// All indices 1-based Int64 (Julia's), as everywhere in this ABI.
#include <algorithm>
#include <numeric>
#include <queue>
#include <vector>

#include "internal.hpp"

using calipso::i64;

namespace {

// symmetric adjacency (0-based, no self loops, sorted, unique) from a CSC pattern of which any triangle(s) may be present
std::vector<std::vector<int>> adjacency(i64 n, const i64* colptr, const i64* rowval) {
    std::vector<std::vector<int>> adj((size_t)n);
    for (i64 c = 0; c < n; ++c)
        for (i64 p = colptr[c] - 1; p < colptr[c + 1] - 1; ++p) {
            const i64 r = rowval[p] - 1;
            if (r == c || r < 0 || r >= n) continue;
            adj[(size_t)r].push_back((int)c); adj[(size_t)c].push_back((int)r);
        }
    for (auto& a : adj) { std::sort(a.begin(), a.end()); a.erase(std::unique(a.begin(), a.end()), a.end()); }
    return adj;
}

// reverse Cuthill-McKee: per connected component a breadth-first numbering from a pseudo-peripheral vertex, neighbours by increasing degree
std::vector<i64> rcm(const std::vector<std::vector<int>>& adj) {
    const int n = (int)adj.size();
    std::vector<int> order; order.reserve(n);
    std::vector<char> seen(n, 0);
    std::vector<int> level(n, 0);
    auto bfs = [&](int root, std::vector<int>& out) {       // returns the vertices of root's component in BFS order, fills `level`
        out.clear(); out.push_back(root);
        std::vector<char> mark(n, 0); mark[root] = 1; level[root] = 0;
        for (size_t h = 0; h < out.size(); ++h) {
            const int v = out[h];
            std::vector<int> nb;
            for (int u : adj[v]) if (!mark[u]) { mark[u] = 1; level[u] = level[v] + 1; nb.push_back(u); }
            std::sort(nb.begin(), nb.end(), [&](int a, int b) { return adj[a].size() != adj[b].size() ? adj[a].size() < adj[b].size() : a < b; });
            out.insert(out.end(), nb.begin(), nb.end());
        }
    };
    std::vector<int> comp, tmp;
    for (int s = 0; s < n; ++s) {
        if (seen[s]) continue;
        int root = s;
        bfs(root, comp);
        for (int iter = 0; iter < 8; ++iter) {              // pseudo-peripheral vertex: farthest vertex of smallest degree, repeat while the depth grows
            const int depth = level[comp.back()];
            int best = comp.back();
            for (int v : comp) if (level[v] == depth && adj[v].size() < adj[best].size()) best = v;
            bfs(best, tmp);
            if (level[tmp.back()] <= depth) break;
            root = best; comp.swap(tmp);
        }
        bfs(root, comp);
        for (int v : comp) seen[v] = 1;
        order.insert(order.end(), comp.begin(), comp.end());
    }
    std::reverse(order.begin(), order.end());
    std::vector<i64> perm((size_t)n);
    for (int k = 0; k < n; ++k) perm[(size_t)k] = order[(size_t)k] + 1;
    return perm;
}

// minimum degree on the quotient graph: an eliminated vertex becomes an element whose members form a clique; the degree of a variable is
// the size of the union of its variable neighbours and the members of its elements (exact external degree), elements reached through the
// pivot are absorbed into the new one.  Ties go to the lowest index.
std::vector<i64> minimum_degree(const std::vector<std::vector<int>>& adj0) {
    const int n = (int)adj0.size();
    std::vector<std::vector<int>> vadj = adj0;             // variable - variable edges still explicit
    std::vector<std::vector<int>> eadj((size_t)n);         // elements adjacent to a variable
    std::vector<std::vector<int>> members((size_t)n);      // members of an element (variables, uneliminated)
    std::vector<char> eliminated(n, 0), absorbed(n, 0);
    std::vector<int> degree(n), mark(n, -1);
    for (int v = 0; v < n; ++v) degree[v] = (int)vadj[v].size();
    typedef std::pair<int, int> DV;
    std::priority_queue<DV, std::vector<DV>, std::greater<DV>> heap;
    for (int v = 0; v < n; ++v) heap.push({degree[v], v});
    std::vector<i64> perm; perm.reserve(n);
    int stamp = 0;
    while ((int)perm.size() < n) {
        DV top = heap.top(); heap.pop();
        const int p = top.second;
        if (eliminated[p] || top.first != degree[p]) continue;
        eliminated[p] = 1; perm.push_back(p + 1);
        // the new element: variable neighbours of p and members of p's elements
        std::vector<int>& Lp = members[p];
        ++stamp; mark[p] = stamp;
        for (int u : vadj[p]) if (!eliminated[u] && mark[u] != stamp) { mark[u] = stamp; Lp.push_back(u); }
        for (int e : eadj[p]) if (!absorbed[e]) {
            for (int u : members[e]) if (!eliminated[u] && mark[u] != stamp) { mark[u] = stamp; Lp.push_back(u); }
            absorbed[e] = 1; members[e].clear(); members[e].shrink_to_fit();
        }
        vadj[p].clear(); eadj[p].clear();
        const int in_new = stamp;
        for (int u : Lp) {
            // prune: edges to members of the new element are now represented by it; absorbed elements go
            std::vector<int>& a = vadj[u];
            a.erase(std::remove_if(a.begin(), a.end(), [&](int w) { return eliminated[w] || mark[w] == in_new; }), a.end());
            std::vector<int>& ea = eadj[u];
            ea.erase(std::remove_if(ea.begin(), ea.end(), [&](int e) { return absorbed[e] != 0; }), ea.end());
            ea.push_back(p);
        }
        for (int u : Lp) {                                  // exact external degrees of the members
            ++stamp; mark[u] = stamp;
            int deg = 0;
            for (int w : vadj[u]) if (mark[w] != stamp) { mark[w] = stamp; ++deg; }
            for (int e : eadj[u]) for (int w : members[e]) if (!eliminated[w] && mark[w] != stamp) { mark[w] = stamp; ++deg; }
            degree[u] = deg;
            heap.push({deg, u});
        }
        // `mark` values of this round are < the next stamp, so the membership test above stays valid only within the round — recompute
        // the membership stamp lazily: nothing outside this block relies on it
    }
    return perm;
}

// nested dissection: the graph is cut by a vertex separator taken from the breadth-first level structure of a pseudo-peripheral vertex (the
// level that halves the vertex count, thinned to the vertices that really touch the far side), the two halves are ordered recursively and the
// separator is eliminated LAST.  The two halves are then independent sub-trees of the elimination tree — for the stage-chained KKT systems of
// trajectory problems (trajectory_optimization/sparsity.jl:28-129) the tree height drops from O(T) stages to O(log T), which is what the
// level-scheduled device factorisation of sparse.hip turns into parallelism (stage-parallel elimination).  Leaves (<= 48
// vertices, or pieces the level structure cannot split) are ordered by minimum degree.
typedef std::vector<std::pair<int, int>> Pieces;     // (first position, count) of every leaf piece / separator, in elimination order
inline void push_pieces(Pieces* pieces, int first, int count) { if (pieces) pieces->push_back({first, count}); }
void nd_recurse(const std::vector<std::vector<int>>& adj, std::vector<int>& verts, std::vector<int>& local, std::vector<int>& level, std::vector<i64>& out, Pieces* pieces) {
    const int m = (int)verts.size();
    auto leaf = [&]() {
        // minimum degree on the induced subgraph
        for (int a = 0; a < m; ++a) local[verts[a]] = a;
        std::vector<std::vector<int>> sub((size_t)m);
        for (int a = 0; a < m; ++a) for (int u : adj[verts[a]]) if (local[u] >= 0) sub[(size_t)a].push_back(local[u]);
        for (int a = 0; a < m; ++a) local[verts[a]] = -1;
        push_pieces(pieces, (int)out.size(), m);
        for (i64 k : minimum_degree(sub)) out.push_back(verts[(size_t)k - 1] + 1);
    };
    if (m <= 48) { leaf(); return; }
    for (int a = 0; a < m; ++a) local[verts[a]] = a;          // membership of the current piece
    // connected components of the piece: independent, no separator needed
    {
        std::vector<int> comp_of((size_t)m, -1); int ncomp = 0;
        std::vector<int> stack;
        for (int a = 0; a < m; ++a) {
            if (comp_of[(size_t)a] >= 0) continue;
            comp_of[(size_t)a] = ncomp; stack.push_back(a);
            while (!stack.empty()) { const int v = stack.back(); stack.pop_back(); for (int u : adj[verts[v]]) { const int lu = local[u]; if (lu >= 0 && comp_of[(size_t)lu] < 0) { comp_of[(size_t)lu] = ncomp; stack.push_back(lu); } } }
            ++ncomp;
        }
        if (ncomp > 1) {
            std::vector<std::vector<int>> parts((size_t)ncomp);
            for (int a = 0; a < m; ++a) parts[(size_t)comp_of[(size_t)a]].push_back(verts[a]);
            for (int a = 0; a < m; ++a) local[verts[a]] = -1;
            for (auto& part : parts) nd_recurse(adj, part, local, level, out, pieces);
            return;
        }
    }
    auto bfs = [&](int root, std::vector<int>& order) {       // level structure of the (connected) piece from root; returns the depth
        for (int v : verts) level[v] = -1;
        order.clear(); order.push_back(root); level[root] = 0;
        for (size_t h = 0; h < order.size(); ++h) { const int v = order[h]; for (int u : adj[v]) if (local[u] >= 0 && level[u] < 0) { level[u] = level[v] + 1; order.push_back(u); } }
        return level[order.back()];
    };
    std::vector<int> order, tmp;
    int root = verts[0], depth = bfs(root, order);
    for (int iter = 0; iter < 6; ++iter) {                    // pseudo-peripheral root: the deepest level structure found
        int best = order.back();
        for (int v : order) if (level[v] == depth && adj[v].size() < adj[best].size()) best = v;
        const int d2 = bfs(best, tmp);
        if (d2 <= depth) { bfs(root, order); break; }
        root = best; depth = d2; order.swap(tmp);
    }
    if (depth < 2) { for (int a = 0; a < m; ++a) local[verts[a]] = -1; leaf(); return; }
    // the level at which half of the vertices have been passed (never the first or the last level)
    std::vector<int> count((size_t)depth + 1, 0);
    for (int v : order) count[(size_t)level[v]] += 1;
    int cut = 1, acc = count[0];
    while (cut < depth - 1 && acc + count[(size_t)cut] < m / 2) { acc += count[(size_t)cut]; ++cut; }
    std::vector<int> A, B, Sep;
    for (int v : order) {
        if (level[v] < cut) A.push_back(v);
        else if (level[v] > cut) B.push_back(v);
        else {
            bool far = false;
            for (int u : adj[v]) if (local[u] >= 0 && level[u] == cut + 1) { far = true; break; }
            (far ? Sep : A).push_back(v);                     // a cut-level vertex without a neighbour beyond the cut belongs to the near half
        }
    }
    for (int a = 0; a < m; ++a) local[verts[a]] = -1;
    if (A.empty() || B.empty()) { leaf(); return; }
    nd_recurse(adj, A, local, level, out, pieces);
    nd_recurse(adj, B, local, level, out, pieces);
    if (!Sep.empty()) {                                       // the separator's own order: minimum degree of its induced subgraph
        const int ms = (int)Sep.size();
        for (int a = 0; a < ms; ++a) local[Sep[a]] = a;
        std::vector<std::vector<int>> sub((size_t)ms);
        for (int a = 0; a < ms; ++a) for (int u : adj[Sep[a]]) if (local[u] >= 0) sub[(size_t)a].push_back(local[u]);
        for (int a = 0; a < ms; ++a) local[Sep[a]] = -1;
        push_pieces(pieces, (int)out.size(), ms);
        for (i64 k : minimum_degree(sub)) out.push_back(Sep[(size_t)k - 1] + 1);
    }
}
std::vector<i64> nested_dissection(const std::vector<std::vector<int>>& adj, Pieces* pieces = nullptr) {
    const int n = (int)adj.size();
    std::vector<int> verts((size_t)n), local((size_t)n, -1), level((size_t)n, -1);
    std::iota(verts.begin(), verts.end(), 0);
    std::vector<i64> out; out.reserve((size_t)n);
    nd_recurse(adj, verts, local, level, out, pieces);
    return out;
}

// permuted upper triangle in the reference's placement order (qdldl.jl:675-737): entries are taken column by column of A and appended to the
// column max(P row, P col) of the result, which leaves the rows inside a column unsorted — the factorisation's operation order follows it
void permute_upper(i64 n, const i64* Ap, const i64* Ai, const i64* iperm, std::vector<i64>& Pp, std::vector<i64>& Pi, std::vector<i64>& map) {
    const i64 nnz = Ap[n] - 1;
    Pp.assign((size_t)n + 1, 0); Pi.assign((size_t)nnz, 0); map.assign((size_t)nnz, 0);
    std::vector<i64> count((size_t)n, 0);
    for (i64 c = 1; c <= n; ++c)
        for (i64 p = Ap[c - 1]; p < Ap[c]; ++p) {
            const i64 r = Ai[p - 1];
            if (r <= c) count[(size_t)std::max(iperm[r - 1], iperm[c - 1]) - 1] += 1;
        }
    Pp[0] = 1;
    for (i64 k = 0; k < n; ++k) Pp[(size_t)k + 1] = Pp[(size_t)k] + count[(size_t)k];
    std::vector<i64> next(Pp.begin(), Pp.end() - 1);
    for (i64 c = 1; c <= n; ++c)
        for (i64 p = Ap[c - 1]; p < Ap[c]; ++p) {
            const i64 r = Ai[p - 1];
            if (r > c) continue;
            const i64 pr = iperm[r - 1], pc = iperm[c - 1];
            const i64 col = std::max(pr, pc);
            const i64 at = next[(size_t)col - 1]++;
            Pi[(size_t)at - 1] = std::min(pr, pc);
            map[(size_t)p - 1] = at;
        }
    const i64 kept = Pp[(size_t)n] - 1;
    Pi.resize((size_t)kept);
}

// elimination tree (Liu's algorithm with path compression) and column counts of L (row-subtree walk on the finished tree) of an upper
// triangular CSC matrix.  Both are unique for a given pattern, so they equal what QDLDL_etree! (qdldl.jl:358-395) computes; the same failure
// cases: -1 for an empty column or an entry below the diagonal.  parent uses -1 for "root" (QDLDL_UNKNOWN).
i64 etree_counts(i64 n, const std::vector<i64>& Pp, const std::vector<i64>& Pi, std::vector<i64>& parent, std::vector<i64>& Lnz) {
    parent.assign((size_t)n, -1); Lnz.assign((size_t)n, 0);
    for (i64 j = 0; j < n; ++j) if (Pp[(size_t)j] == Pp[(size_t)j + 1]) return -1;
    std::vector<i64> anc((size_t)n, -1);
    for (i64 j = 0; j < n; ++j)
        for (i64 p = Pp[(size_t)j] - 1; p < Pp[(size_t)j + 1] - 1; ++p) {
            i64 i = Pi[(size_t)p] - 1;
            if (i > j) return -1;
            while (i != -1 && i < j) {
                const i64 nxt = anc[(size_t)i];
                anc[(size_t)i] = j;
                if (nxt == -1) parent[(size_t)i] = j;
                i = nxt;
            }
        }
    std::vector<i64> mark((size_t)n, -1);
    for (i64 j = 0; j < n; ++j) {
        mark[(size_t)j] = j;
        for (i64 p = Pp[(size_t)j] - 1; p < Pp[(size_t)j + 1] - 1; ++p)
            for (i64 i = Pi[(size_t)p] - 1; mark[(size_t)i] != j; i = parent[(size_t)i]) { mark[(size_t)i] = j; Lnz[(size_t)i] += 1; }
    }
    return std::accumulate(Lnz.begin(), Lnz.end(), (i64)0);
}

bool is_permutation(i64 n, const i64* perm) {
    std::vector<char> seen((size_t)n, 0);
    for (i64 k = 0; k < n; ++k) { const i64 v = perm[k]; if (v < 1 || v > n || seen[(size_t)v - 1]) return false; seen[(size_t)v - 1] = 1; }
    return true;
}

}  // namespace

namespace calipso {
// nested dissection with its pieces (leaf pieces <= 48 vertices and separators, each contiguous in the order): the supernodes of the multifrontal
// factorisation of sparse.hip.  perm: 1-based, n entries; pieces: (first position 0-based, count)
int nested_dissection_pieces(i64 n, const i64* colptr, const i64* rowval, i64* perm, std::vector<std::pair<int, int>>& pieces) {
    if (n < 0 || !colptr || !perm || (!rowval && colptr[n] > 1)) return CALIPSO_ERR_ARGUMENT;
    pieces.clear();
    const std::vector<i64> p = nested_dissection(adjacency(n, colptr, rowval), &pieces);
    std::copy(p.begin(), p.end(), perm);
    return CALIPSO_OK;
}
// one gate for every entry point that walks a caller's CSC pattern: colptr[0] == 1, non-decreasing, fewer than 2^31 entries, 1 <= rowval <= n
bool csc_pattern_ok(i64 n, const i64* colptr, const i64* rowval) {
    if (n < 0 || !colptr || colptr[0] != 1) return false;
    for (i64 c = 0; c < n; ++c) if (colptr[c + 1] < colptr[c]) return false;
    const i64 nnz = colptr[n] - 1;
    if (nnz < 0 || nnz >= ((i64)1 << 31) || (nnz > 0 && !rowval)) return false;
    for (i64 p = 0; p < nnz; ++p) if (rowval[p] < 1 || rowval[p] > n) return false;
    return true;
}
}  // namespace calipso

extern "C" {

// method 0: natural (1..n), 1: reverse Cuthill-McKee, 2: minimum degree, 4: nested dissection (3 is "the caller's order" in the analyse calls).  Pattern: CSC, 1-based, any triangle(s).  perm[k] = the vertex eliminated k-th.
int32_t calipso_hip_ordering(int64_t n, const int64_t* colptr, const int64_t* rowval, int32_t method, int64_t* perm) {
    if (n < 0 || !colptr || !perm || method < 0 || method > 4 || method == 3 || !calipso::csc_pattern_ok(n, colptr, rowval)) return CALIPSO_ERR_ARGUMENT;
    if (method == 0) { for (i64 k = 0; k < n; ++k) perm[k] = k + 1; return CALIPSO_OK; }
    const auto adj = adjacency(n, colptr, rowval);
    const std::vector<i64> p = method == 1 ? rcm(adj) : method == 2 ? minimum_degree(adj) : nested_dissection(adj);
    std::copy(p.begin(), p.end(), perm);
    return CALIPSO_OK;
}

// Symbolic phase for the upper triangle of A under `perm` (NULL = natural): Pp[n+1], Pi[nnz(triu A)], AtoPAPt[nnz A] (0 for entries below the
// diagonal), etree[n] (-1 = root), Lnz[n]; any output may be NULL.  Returns sum(Lnz) = nnz(L), -1 in QDLDL_etree!'s failure cases, or a negative
// status < -1.  info (may be NULL): [0] half bandwidth of P A P', [1] nnz(triu A).
int64_t calipso_hip_symbolic(int64_t n, const int64_t* colptr, const int64_t* rowval, const int64_t* perm, int64_t* Pp, int64_t* Pi, int64_t* AtoPAPt,
                             int64_t* etree, int64_t* Lnz, int64_t info[2]) {
    if (n < 1 || !calipso::csc_pattern_ok(n, colptr, rowval)) return CALIPSO_ERR_ARGUMENT;
    std::vector<i64> iperm((size_t)n);
    if (perm) { if (!is_permutation(n, perm)) return CALIPSO_ERR_ARGUMENT; for (i64 k = 0; k < n; ++k) iperm[(size_t)perm[k] - 1] = k + 1; }   // invperm (qdldl.jl:143)
    else for (i64 k = 0; k < n; ++k) iperm[(size_t)k] = k + 1;
    std::vector<i64> pp, pi, map, parent, lnz;
    permute_upper(n, colptr, rowval, iperm.data(), pp, pi, map);
    const i64 total = etree_counts(n, pp, pi, parent, lnz);
    i64 hb = 0;
    for (i64 j = 0; j < n; ++j) for (i64 p = pp[(size_t)j] - 1; p < pp[(size_t)j + 1] - 1; ++p) hb = std::max(hb, j + 1 - pi[(size_t)p]);
    if (Pp) std::copy(pp.begin(), pp.end(), Pp);
    if (Pi) std::copy(pi.begin(), pi.end(), Pi);
    if (AtoPAPt) std::copy(map.begin(), map.end(), AtoPAPt);
    if (etree) for (i64 k = 0; k < n; ++k) etree[k] = parent[(size_t)k] < 0 ? -1 : parent[(size_t)k] + 1;
    if (Lnz) std::copy(lnz.begin(), lnz.end(), Lnz);
    if (info) { info[0] = hb; info[1] = (i64)pi.size(); }
    return total;
}

}  // extern "C"
