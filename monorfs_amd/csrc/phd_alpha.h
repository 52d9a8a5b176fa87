// phd_alpha.h — k_weight_alpha and the association solvers it uses.
#pragma once
#include "phd_device.h"

// =================================================================================================
// k_weight_alpha — WeightAlpha (PHDNavigator.cs:373-393)
// =================================================================================================

// Permutations of LexicographicalPairing (GraphCombinatorics.cs:280-334) on n <= 5 entries, held as 4-bit
// fields of one register (entry i in bits [4i, 4i+4)) so that nothing is indexed dynamically in memory.
__device__ __forceinline__ int pk_get(unsigned int perm, int i) { return (int) ((perm >> (4 * i)) & 15u); }

__device__ __forceinline__ unsigned int pk_set(unsigned int perm, int i, int v)
{
	return (perm & ~(15u << (4 * i))) | ((unsigned int) v << (4 * i));
}

__device__ __forceinline__ unsigned int pk_reverse(unsigned int perm, int from, int to)   // Array.Reverse on [from, to)
{
	for (int i = from, j = to - 1; i < j; i++, j--) {
		int vi = pk_get(perm, i), vj = pk_get(perm, j);
		perm = pk_set(pk_set(perm, i, vj), j, vi);
	}
	return perm;
}

__device__ __forceinline__ bool pk_last(unsigned int perm, int n)   // lastpermutation, :341-350
{
	for (int i = 1; i < n; i++) {
		if (pk_get(perm, i - 1) < pk_get(perm, i)) return false;
	}
	return true;
}

__device__ __forceinline__ unsigned int pk_next(unsigned int perm, int n, int measurestart)   // :306-331
{
	int x, y;
	for (x = n - 2; x > 0; x--) {
		if (pk_get(perm, x) < pk_get(perm, x + 1)) break;
	}
	for (y = n - 1; y > x; y--) {
		if (pk_get(perm, x) < pk_get(perm, y)) break;
	}
	int vx = pk_get(perm, x), vy = pk_get(perm, y);
	perm = pk_set(pk_set(perm, x, vy), y, vx);
	perm = pk_reverse(perm, x + 1, n);
	return pk_reverse(perm, measurestart, n);
}

// the m-th permutation of 0 .. n - 1 (n <= 5) in lexicographic order, the identity being number 0: what pk_next reaches from
// the identity in m steps when measurestart == n
__device__ __forceinline__ unsigned int pk_unrank(int m, int n)
{
	unsigned int avail = 0x43210u, perm = 0x43210u & ~((1u << (4 * n)) - 1u);
	int fct = 1;
	for (int i = 2; i < n; i++) fct *= i;   // (n - 1)!
	for (int i = 0; i < n; i++) {
		int d = 0;
		while (m >= fct) { m -= fct; d++; }
		perm |= ((avail >> (4 * d)) & 15u) << (4 * i);
		const unsigned int low = (1u << (4 * d)) - 1u;
		avail = (avail & low) | ((avail >> 4) & ~low);
		if (n - 1 - i > 1) fct /= (n - 1 - i);
	}
	return perm;
}

// the first n (<= 5) 12-bit fields of v in ascending order (9-comparator network)
__device__ __forceinline__ unsigned long long pk_sort5(unsigned long long v, int n)
{
	int f[5];
#pragma unroll
	for (int i = 0; i < 5; i++) f[i] = (i < n) ? (int) ((v >> (12 * i)) & 4095) : 4096 + i;
#define PK_CE(a_, b_) { int lo_ = min(f[a_], f[b_]), hi_ = max(f[a_], f[b_]); f[a_] = lo_; f[b_] = hi_; }
	PK_CE(0, 1) PK_CE(3, 4) PK_CE(2, 4) PK_CE(2, 3) PK_CE(1, 4) PK_CE(0, 3) PK_CE(0, 2) PK_CE(1, 3) PK_CE(1, 2)
#undef PK_CE
	unsigned long long o = 0;
#pragma unroll
	for (int i = 0; i < 5; i++) o |= (unsigned long long) (f[i] & 4095) << (12 * i);
	return o;
}

// log-sum-exp over every pairing of a cluster with n <= 5 rows, enumerated exactly like
// LexicographicalPairing(component, map.Count) (`modelsize` is compared with COMPACTED row indices,
// PHDNavigator.cs:493 / GraphCombinatorics.cs:293-299). mat: n x n, row stride 5, stride `ms` between entries.
__device__ double cluster_enumerate(const double* mat, int ms, int n, int modelsize, double* rec = nullptr)
{
	int measurestart = n;
	for (int i = 0; i < n; i++) {
		if (i >= modelsize) { measurestart = i; break; }
	}
	unsigned int first = 0x43210u;   // row keys, sorted
	first = pk_reverse(first, measurestart, n);

	double mx = -INFINITY, value = 0;
	for (int pass = 0; pass < 2; pass++) {
		unsigned int perm = first;
		int m = 0;
		for (;;) {
			double v = 0;
			for (int i = 0; i < n; i++) v += mat[(i * 5 + pk_get(perm, i)) * ms];   // AssignmentValue
			if (pass == 0) {
				mx = fmax(mx, v);
				if (rec) rec[m] = v;   // logcomp[m] = assignment.Item2 (PHDNavigator.cs:507)
			}
			else value += exp(v - mx);
			m++;
			if (pk_last(perm, n)) break;
			perm = pk_next(perm, n, measurestart);
		}
		if (pass == 0 && isinf(mx) && mx < 0) return -INFINITY;   // LogSumExp, MatrixExtensions.cs:379-381
	}
	return mx + log(value);
}

// ---- clusters with more than 5 rows: MurtyPairing (GraphCombinatorics.cs:241-272) run by one wave ----
// Lane l owns rows and columns l, l + 64, ... (NT of each, in registers): clusters of up to 64 rows (NT = 1) work in
// the particle's own workspace, larger ones (NT = 2: <= 128 rows, NT = 4: <= 256) in a block taken from the handle's
// association slab (StepBufs::bigws). The reference has no size limit; a 256-row cluster costs it 200 x 255 Hungarian
// solutions of 256^3 steps each — beyond that the step fails with PHD_ERR_ASSOCIATION (the state is kept).
#define MURTY_NMAX  64    // rows of a cluster solved in the per-particle workspace
#define MURTY_NBIG  256   // rows of the largest cluster solved at all (column indices are bytes)
#define MURTY_OUT   200   // logcomp.Length (PHDNavigator.cs:469)
#define MURTY_POOL  208   // frontier (<= 201 live entries) + the node being expanded + its child
#define MURTY_ELCAP 208   // eliminated edges a node can hold: one per generation, at most MURTY_OUT generations

struct MurtyNodes {       // per-particle workspace in HBM, touched only when a cluster of more than 5 rows exists
	unsigned char      asg[MURTY_POOL * MURTY_NMAX];     // assignment (row -> column) of every node
	unsigned long long forced[MURTY_POOL];               // forced rows (bit = row); a forced edge is (row, asg[row])
	unsigned short     elist[MURTY_POOL * MURTY_ELCAP];  // eliminated edges, row << 8 | column
	int                nelim[MURTY_POOL];
	double             profit[MURTY_NMAX * MURTY_NMAX];  // the cluster's matrix
	double             reduced[MURTY_NMAX * MURTY_NMAX]; // and the copy a child node solves on
	double             dvec[MURTY_OUT][6];               // gradient mode: dlogcompdp of the enumerated pairings (PHDNavigator.cs:675)
	double             jp[MURTY_NMAX * 18];              // gradient mode: MeasurementJacobianP of the cluster's landmarks
	int                L[MURTY_NMAX], Z[MURTY_NMAX];     // the cluster's landmarks / measurements (test surface: unused)
};

// The same arrays for a cluster of n rows, wherever they live (`stride` = row capacity of asg).
struct MurtyWs {
	unsigned char*      asg;
	unsigned long long* forced;    // [POOL][NT]
	unsigned short*     elist;
	int*                nelim;
	double*             profit;
	double*             reduced;
	double*             jp;
	int*                L;
	int*                Z;
	int                 stride;
};

__device__ __forceinline__ MurtyWs murty_ws_particle(MurtyNodes* nd)
{
	MurtyWs w;
	w.asg = nd->asg; w.forced = nd->forced; w.elist = nd->elist; w.nelim = nd->nelim;
	w.profit = nd->profit; w.reduced = nd->reduced; w.jp = nd->jp; w.L = nd->L; w.Z = nd->Z;
	w.stride = MURTY_NMAX;
	return w;
}

// bytes of a slab block for a cluster of n rows (multiple of 16)
__host__ __device__ inline size_t murty_big_bytes(int n)
{
	const size_t nt = (n <= 64) ? 1 : ((n <= 128) ? 2 : 4);   // forced-row words per node: the NT of wave_murty_any
	size_t b = 0;
	b += 2 * (size_t) n * n * 8;                 // profit, reduced
	b += (size_t) n * 18 * 8;                    // jp
	b += (size_t) MURTY_POOL * nt * 8;           // forced
	b += (size_t) MURTY_POOL * 4;                // nelim
	b += 2 * (size_t) n * 4;                     // L, Z
	b += (size_t) MURTY_POOL * MURTY_ELCAP * 2;  // elist
	b += (size_t) MURTY_POOL * n;                // asg
	return (b + 63) & ~(size_t) 63;
}

__device__ __forceinline__ MurtyWs murty_ws_carve(char* base, int n)
{
	const size_t nt = (n <= 64) ? 1 : ((n <= 128) ? 2 : 4);
	MurtyWs w;
	char* q = base;
	w.profit  = (double*) q; q += (size_t) n * n * 8;
	w.reduced = (double*) q; q += (size_t) n * n * 8;
	w.jp      = (double*) q; q += (size_t) n * 18 * 8;
	w.forced  = (unsigned long long*) q; q += (size_t) MURTY_POOL * nt * 8;
	w.nelim   = (int*) q; q += (size_t) MURTY_POOL * 4;
	w.L       = (int*) q; q += (size_t) n * 4;
	w.Z       = (int*) q; q += (size_t) n * 4;
	w.elist   = (unsigned short*) q; q += (size_t) MURTY_POOL * MURTY_ELCAP * 2;
	w.asg     = (unsigned char*) q;
	w.stride  = n;
	return w;
}

__device__ __forceinline__ double wave_min(double v)
{
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
	return v;
}


// entry `idx` (wave-uniform) of an array spread as element lane + 64 t in x[t]
template <int NT>
__device__ __forceinline__ double spread_get(const double (&x)[NT], int idx)
{
	double v = x[0];
#pragma unroll
	for (int t = 1; t < NT; t++) v = ((idx >> 6) == t) ? x[t] : v;
	return __shfl(v, idx & 63, 64);
}

template <int NT>
__device__ __forceinline__ int spread_get(const int (&x)[NT], int idx)
{
	int v = x[0];
#pragma unroll
	for (int t = 1; t < NT; t++) v = ((idx >> 6) == t) ? x[t] : v;
	return __shfl(v, idx & 63, 64);
}

template <int NT>
__device__ __forceinline__ void spread_set(int (&x)[NT], int idx, int val, int lane)
{
#pragma unroll
	for (int t = 0; t < NT; t++) {
		if (lane == (idx & 63) && (idx >> 6) == t) x[t] = val;
	}
}

// Hungarian (GraphCombinatorics.cs:64-175) with lane l owning rows l + 64 t (labelx, matchx, visitx) and columns l + 64 t
// (labely, matchy, visity, slack, parent). Every arithmetic step is the serial algorithm's, the argmin keeps its
// first-minimum tie-break (smallest index: t first, then the lane), so the assignment is the reference's.
template <int NT>
__device__ bool wave_hungarian(const double* mat, int n, int lane, int (&matchx_out)[NT])
{
	double labelx[NT], labely[NT], slack[NT];
	int    matchx[NT], matchy[NT], parent[NT];
	bool   act[NT];
#pragma unroll
	for (int t = 0; t < NT; t++) {
		const int i = lane + 64 * t;
		act[t] = i < n;
		double f = 0;   // FoldRows(Math.Max, 0)
		if (act[t]) {
			for (int k = 0; k < n; k++) f = fmax(f, mat[i * n + k]);
		}
		labelx[t] = f; labely[t] = 0; slack[t] = INFINITY;
		matchx[t] = -1; matchy[t] = -1; parent[t] = 0;
	}
	for (;;) {
		int root = -1;   // Array.IndexOf(matchx, -1)
#pragma unroll
		for (int t = 0; t < NT; t++) {
			const unsigned long long um = ballot64(act[t] && matchx[t] == -1);
			if (root < 0 && um) root = 64 * t + __ffsll((long long) um) - 1;
		}
		if (root < 0) break;
		const double lxr = spread_get<NT>(labelx, root);
		unsigned int visitx = 0, visity = 0;   // bit t <-> row / column lane + 64 t
#pragma unroll
		for (int t = 0; t < NT; t++) {
			parent[t] = root;
			slack[t] = act[t] ? (lxr + labely[t] - mat[root * n + lane + 64 * t]) : INFINITY;
			if (lane == (root & 63) && (root >> 6) == t) visitx |= 1u << t;
		}
		int imin = 0;
		for (;;) {
			double val = INFINITY;
#pragma unroll
			for (int t = 0; t < NT; t++) {
				if (act[t] && !((visity >> t) & 1u) && slack[t] < val) val = slack[t];
			}
			const double delta = wave_min(val);
			if (isinf(delta) && delta > 0) return false;   // no solution
			imin = -1;
#pragma unroll
			for (int t = 0; t < NT; t++) {
				const unsigned long long bal = ballot64(act[t] && !((visity >> t) & 1u) && slack[t] == delta);
				if (imin < 0 && bal) imin = 64 * t + __ffsll((long long) bal) - 1;
			}
#pragma unroll
			for (int t = 0; t < NT; t++) {
				if ((visitx >> t) & 1u) labelx[t] -= delta;
				if (act[t]) {
					if ((visity >> t) & 1u) labely[t] += delta;
					else slack[t] -= delta;
				}
			}
			if (lane == (imin & 63)) visity |= 1u << (imin >> 6);
			const int my = spread_get<NT>(matchy, imin);
			if (my == -1) break;
			if (lane == (my & 63)) visitx |= 1u << (my >> 6);
			const double lxm = spread_get<NT>(labelx, my);
#pragma unroll
			for (int t = 0; t < NT; t++) {
				if (act[t] && !((visity >> t) & 1u)) {
					const double md = lxm + labely[t] - mat[my * n + lane + 64 * t];
					if (md < slack[t]) { slack[t] = md; parent[t] = my; }
				}
			}
		}
		int py = imin, px = spread_get<NT>(parent, py);
		while (px != root) {
			const int ty = spread_get<NT>(matchx, px);
			spread_set<NT>(matchx, px, py, lane);
			spread_set<NT>(matchy, py, px, lane);
			py = ty;
			px = spread_get<NT>(parent, py);
		}
		spread_set<NT>(matchx, px, py, lane);
		spread_set<NT>(matchy, py, px, lane);
	}
#pragma unroll
	for (int t = 0; t < NT; t++) matchx_out[t] = matchx[t];
	return true;
}

// AssignmentValue (GraphCombinatorics.cs:183-197), summed in row order
template <int NT>
__device__ __forceinline__ double wave_assignment_value(const double* profit, int n, const int (&matchx)[NT])
{
	double total = 0;
	for (int i = 0; i < n; i++) total += profit[i * n + spread_get<NT>(matchx, i)];
	return total;
}

// LDS scratch of the Murty path (the matrices and the nodes live in HBM: such clusters are rare, LDS is better spent
// on occupancy)
struct MurtyLds {
	double* logcomp;   // [MURTY_OUT]
	double* fkey;      // [MURTY_POOL] frontier priorities, ascending
	int*    fnode;     // [MURTY_POOL] frontier node slots
	int*    freelist;  // [MURTY_POOL]
};
#define MURTY_LDS_DOUBLES (MURTY_OUT + MURTY_POOL + (2 * MURTY_POOL + 1) / 2)

__device__ __forceinline__ MurtyLds murty_lds(double* base)
{
	MurtyLds ws;
	ws.logcomp  = base;
	ws.fkey     = ws.logcomp + MURTY_OUT;
	ws.fnode    = (int*) (ws.fkey + MURTY_POOL);
	ws.freelist = ws.fnode + MURTY_POOL;
	return ws;
}

// Enumerate the pairings of one cluster best-first and record their values into logcomp exactly as
// the loop of SetLogLikelihood does (PHDNavigator.cs:501-509), including its read of the stale
// logcomp[m] left by earlier clusters. Returns the number of values written. Wave-uniform.
// hook(m, unsolved, colof): the pairing recorded at index m; colof(row) = its column (wave-uniform row).
// (TAG: kernels with different register budgets — the one-launch chain against the four-workgroups-per-CU kernels — must
// not share one out-of-line copy: it would be compiled for the larger budget and drag the others' occupancy down)
template <int NT, int TAG, class Hook>
__device__ __noinline__ int wave_murty(const MurtyLds& ws, const MurtyWs& nd, int n, int lane, Hook hook)
{
	const double* profit = nd.profit;
	int nfree = 0;
	if (lane == 0) {
		for (int i = 0; i < MURTY_POOL; i++) ws.freelist[i] = MURTY_POOL - 1 - i;
	}
	nfree = MURTY_POOL;
	lds_fence();
	int fsize = 0, m = 0;

	// frontier.Add(priority, node): keep ascending order, a new entry goes after its equals
	// (PriorityQueue.Add re-sorts the list, GraphCombinatorics.cs:638-642; canonical stable order);
	// only the best (MURTY_OUT + 1 - m) entries can ever be popped, the rest is dropped.
	auto frontier_add = [&](double key, int slot) {
		int cap = MURTY_OUT + 1 - m;
		int pos = 0;
		for (int b = 0; b < MURTY_POOL; b += 64) {
			int idx = b + lane;
			pos += __popcll(ballot64(idx < fsize && ws.fkey[idx] <= key));
		}
		if (fsize >= cap && pos == 0) {   // would be the worst of a full frontier
			if (lane == 0) ws.freelist[nfree] = slot;
			nfree++;
			lds_fence();
			return;
		}
		// shift [pos, fsize) up by one
		double kreg[4]; int nreg[4];
#pragma unroll
		for (int u = 0; u < 4; u++) {
			int idx = u * 64 + lane;
			kreg[u] = (idx < fsize) ? ws.fkey[idx] : 0.0;
			nreg[u] = (idx < fsize) ? ws.fnode[idx] : 0;
		}
		lds_fence();
#pragma unroll
		for (int u = 0; u < 4; u++) {
			int idx = u * 64 + lane;
			if (idx >= pos && idx < fsize) { ws.fkey[idx + 1] = kreg[u]; ws.fnode[idx + 1] = nreg[u]; }
		}
		if (lane == 0) { ws.fkey[pos] = key; ws.fnode[pos] = slot; }
		fsize++;
		lds_fence();
		if (fsize > cap) {   // drop the worst (front)
			int dropped = ws.fnode[0];
#pragma unroll
			for (int u = 0; u < 4; u++) {
				int idx = u * 64 + lane;
				kreg[u] = (idx < fsize) ? ws.fkey[idx] : 0.0;
				nreg[u] = (idx < fsize) ? ws.fnode[idx] : 0;
			}
			lds_fence();
#pragma unroll
			for (int u = 0; u < 4; u++) {
				int idx = u * 64 + lane;
				if (idx >= 1 && idx < fsize) { ws.fkey[idx - 1] = kreg[u]; ws.fnode[idx - 1] = nreg[u]; }
			}
			fsize--;
			if (lane == 0) ws.freelist[nfree] = dropped & 0xffff;
			nfree++;
			lds_fence();
		}
	};
	auto alloc = [&]() {
		nfree--;
		return ws.freelist[nfree];
	};
	// node `slot` <- assignment mx, forced rows (bit t of `fr` <-> row lane + 64 t), eliminated edges: those of node
	// `parent` (none when < 0) and (er, ec) when er >= 0
	auto store_node = [&](int slot, const int (&mx)[NT], unsigned int fr, int parent, int er, int ec) {
#pragma unroll
		for (int t = 0; t < NT; t++) {
			const int i = lane + 64 * t;
			if (i < n) nd.asg[(size_t) slot * nd.stride + i] = (unsigned char) mx[t];
			const unsigned long long w = ballot64((fr >> t) & 1u);
			if (lane == 0) nd.forced[slot * NT + t] = w;
		}
		int ne = 0;
		if (parent >= 0) {
			ne = nd.nelim[parent];
			for (int e = lane; e < ne; e += 64) nd.elist[slot * MURTY_ELCAP + e] = nd.elist[parent * MURTY_ELCAP + e];
		}
		if (lane == 0) {
			if (er >= 0 && ne < MURTY_ELCAP) nd.elist[slot * MURTY_ELCAP + ne] = (unsigned short) ((er << 8) | ec);
			nd.nelim[slot] = ne + ((er >= 0 && ne < MURTY_ELCAP) ? 1 : 0);
		}
		__threadfence_block();
	};

	// first node: no forced, no eliminated edges
	{
		int slot = alloc();
		int mx[NT];
		bool solved = wave_hungarian<NT>(profit, n, lane, mx);
		if (!solved) {
#pragma unroll
			for (int t = 0; t < NT; t++) mx[t] = 0;
		}
		store_node(slot, mx, 0u, -1, -1, 0);
		double value = solved ? wave_assignment_value<NT>(profit, n, mx) : -INFINITY;
		// an unsolved first node is yielded with value -inf and has no children (GraphCombinatorics.cs:245-249,473)
		frontier_add(value, solved ? slot : (slot | 0x10000));
	}

	while (fsize > 0) {
		// best = frontier.Pop(out value)
		const double value = ws.fkey[fsize - 1];
		const int    code  = ws.fnode[fsize - 1];
		fsize--;
		// foreach body of SetLogLikelihood (PHDNavigator.cs:502-509)
		if (m >= MURTY_OUT || (ws.logcomp[m] - ws.logcomp[0] < -10)) break;
		if (lane == 0) ws.logcomp[m] = value;
		const int slot = code & 0xffff;
		hook(m, (code & 0x10000) != 0, [&](int row) { return (int) nd.asg[(size_t) slot * nd.stride + row]; });
		m++;
		lds_fence();
		if (code & 0x10000) continue;   // unsolved: no children
		// children (MurtyNode.Children, GraphCombinatorics.cs:469-509)
		int pa[NT];
		unsigned int fr = 0, fc = 0;             // forced rows / forced columns of the child being built (bit t <-> lane + 64 t)
		unsigned long long rem[NT];              // rows the parent does not force (wave-uniform)
		int R = 0;
#pragma unroll
		for (int t = 0; t < NT; t++) {
			const int i = lane + 64 * t;
			pa[t] = (i < n) ? nd.asg[(size_t) slot * nd.stride + i] : 0;
			const unsigned long long fw = nd.forced[slot * NT + t];
			if ((fw >> lane) & 1ull) fr |= 1u << t;
			const int left = n - 64 * t;
			const unsigned long long rows = (left >= 64) ? ~0ull : ((left <= 0) ? 0ull : ((1ull << left) - 1ull));
			rem[t] = ~fw & rows;
			R += __popcll(rem[t]);
		}
		// forced columns = columns of the forced rows (reduceprofit, GraphCombinatorics.cs:206-234)
#pragma unroll
		for (int t = 0; t < NT; t++) {
			unsigned long long fw = ballot64((fr >> t) & 1u);
			while (fw) {
				const int l = __ffsll((long long) fw) - 1;
				fw &= fw - 1;
				const int col = __shfl(pa[t], l, 64);
				if (lane == (col & 63)) fc |= 1u << (col >> 6);
			}
		}
		const int pne = nd.nelim[slot];
		for (int c = 0; c < R - 1; c++) {
			int er = -1;   // remaining[c]
#pragma unroll
			for (int t = 0; t < NT; t++) {
				if (er < 0 && rem[t]) {
					er = 64 * t + __ffsll((long long) rem[t]) - 1;
					rem[t] &= rem[t] - 1;
				}
			}
			const int ec = spread_get<NT>(pa, er);
			// reduceprofit: a forced row keeps 1 on its edge and nothing else, the other rows lose the forced columns
			for (int i = 0; i < n; i++) {
				const int  pai = spread_get<NT>(pa, i);
				const bool fi  = __shfl((int) fr, i & 63, 64) & (1 << (i >> 6));
#pragma unroll
				for (int t = 0; t < NT; t++) {
					const int k = lane + 64 * t;
					if (k < n) {
						double v = profit[i * n + k];
						if (fi) v = (k == pai) ? 1.0 : -INFINITY;
						else if ((fc >> t) & 1u) v = -INFINITY;
						nd.reduced[i * n + k] = v;
					}
				}
			}
			__threadfence_block();
			// eliminated edges: the parent's and (er, ec)
			for (int e = lane; e < pne; e += 64) {
				const int ed = nd.elist[slot * MURTY_ELCAP + e];
				nd.reduced[(ed >> 8) * n + (ed & 255)] = -INFINITY;
			}
			if (lane == 0) nd.reduced[er * n + ec] = -INFINITY;
			__threadfence_block();
			int mx[NT];
			const bool solved = wave_hungarian<NT>(nd.reduced, n, lane, mx);
			if (solved) {
				const int cs = alloc();
				store_node(cs, mx, fr, slot, er, ec);
				frontier_add(wave_assignment_value<NT>(profit, n, mx), cs);
			}
			// the next child also forces (er, ec)
			if (lane == (er & 63)) fr |= 1u << (er >> 6);
			if (lane == (ec & 63)) fc |= 1u << (ec >> 6);
		}
		if (lane == 0) ws.freelist[nfree] = slot;
		nfree++;
		lds_fence();
	}
	return m;
}

// the same for any cluster size the solver takes: the number of rows picks the register layout
template <int TAG, class Hook>
__device__ __forceinline__ int wave_murty_any(const MurtyLds& ws, const MurtyWs& nd, int n, int lane, Hook hook)
{
	if (n <= 64) return wave_murty<1, TAG>(ws, nd, n, lane, hook);
	if (n <= 128) return wave_murty<2, TAG>(ws, nd, n, lane, hook);
	return wave_murty<4, TAG>(ws, nd, n, lane, hook);
}

// ---- gradient mode of QuasiSetLogLikelihood (PHDNavigator.cs:543-713 with calcgradient) ----
// MeasurementJacobianP (PRM3DMeasurer.cs:185-211): jprojection (3 x 3) times [-R(q*) | -R(q*) [m - t]x], row-major 3 x 6
__device__ void jacobian_p(const DevParams& prm, const PoseD& pose, const double m[3], double* Jp)
{
	if (prm.linear2d) {   // Linear2DMeasurer.MeasurementJacobianP (Linear2DMeasurer.cs:133-137): -I on (x, y)
		for (int i = 0; i < 18; i++) Jp[i] = 0.0;
		Jp[0] = -1.0; Jp[7] = -1.0;
		return;
	}
	const double diff[3] = {m[0] - pose.t[0], m[1] - pose.t[1], m[2] - pose.t[2]};
	double l[3];
	to_local(pose, diff, l);
	const double f = prm.focal;
	const double mag = ((l[2] > 0) ? 1.0 : -1.0) * sqrt(l[0] * l[0] + l[1] * l[1] + l[2] * l[2]);
	const double jp[9] = {f / l[2], 0, -f * l[0] / (l[2] * l[2]),  0, f / l[2], -f * l[1] / (l[2] * l[2]),
	                      l[0] / mag, l[1] / mag, l[2] / mag};
	double rc[9], jlocal[18];
	conj_matrix(pose, rc);
	const double cross[9] = {0, -diff[2], diff[1],  diff[2], 0, -diff[0],  -diff[1], diff[0], 0};
	// (every index static: the small matrices stay in registers — left to the compiler's own unrolling they went to scratch)
#pragma unroll
	for (int i = 0; i < 3; i++) {
#pragma unroll
		for (int j = 0; j < 3; j++) {
			double acc = 0;
#pragma unroll
			for (int k = 0; k < 3; k++) acc += (-1.0 * rc[i * 3 + k]) * cross[k * 3 + j];
			jlocal[i * 6 + j]     = -1.0 * rc[i * 3 + j];
			jlocal[i * 6 + 3 + j] = acc;
		}
	}
#pragma unroll
	for (int i = 0; i < 3; i++) {
#pragma unroll
		for (int j = 0; j < 6; j++) {
			double acc = 0;
#pragma unroll
			for (int k = 0; k < 3; k++) acc += jp[i * 3 + k] * jlocal[k * 6 + j];
			Jp[i * 6 + j] = acc;
		}
	}
}

// component t of dlldp[i, k] = (z_k - zhat_i)' R^-1 Jp_i (:605-608)
__device__ __forceinline__ double pair_gradient(const DevParams& prm, const double* nu, const double* Jp, int t)
{
	double acc = 0;
#pragma unroll
	for (int b = 0; b < 3; b++) {
		double u = 0;
#pragma unroll
		for (int e = 0; e < 3; e++) u += nu[e] * prm.Rinv[e * 3 + b];
		acc += u * Jp[b * 6 + t];
	}
	return acc;
}

// The pairings of a cluster with n <= 5 rows in the order of LexicographicalPairing (GraphCombinatorics.cs:280-334),
// by a whole wave: every lane walks the same permutations and computes the same value (AssignmentValue, rows in order);
// `hook(m, perm)` lets the lanes add what they keep per pairing. mat: n x n, row stride 5. Returns the count.
template <class Hook>
__device__ int cluster_enumerate_wave(const double* mat, int n, int modelsize, double* logcomp, int lane, Hook hook)
{
	int measurestart = n;
	for (int i = 0; i < n; i++) {
		if (i >= modelsize) { measurestart = i; break; }
	}
	unsigned int perm = pk_reverse(0x43210u, measurestart, n);
	int m = 0;
	for (;;) {
		double v = 0;
		for (int i = 0; i < n; i++) v += mat[i * 5 + pk_get(perm, i)];
		if (m < MURTY_OUT) {                       // m >= logcomp.Length ends the loop (:672); 5! = 120 never gets there
			if (lane == 0) logcomp[m] = v;
			hook(m, perm);
			m++;
		}
		if (pk_last(perm, n)) break;
		perm = pk_next(perm, n, measurestart);
	}
	lds_fence();
	return m;
}

// Test surface (phd_test_pairing): the pairing enumerators on a matrix handed in by the host, so that the vectors of
// GraphCombinatoricsTest.cs reach the device code itself. One wave. mode 0: MurtyPairing (best first; the cut of
// SetLogLikelihood's loop is kept out by pre-filling logcomp), mode 1: LexicographicalPairing(matrix, modelsize), n <= 5.
__global__ __launch_bounds__(64) void k_test_pairing(MurtyNodes* nodes, char* bigws, const double* matrix, int n, int mode, int modelsize,
                                                     int maxcount, int* assignments, double* values, int* count)
{
	__shared__ double lds[MURTY_LDS_DOUBLES + 2];
	const int lane = threadIdx.x;
	const MurtyLds ws = murty_lds(lds);
	const MurtyWs nd = (n <= MURTY_NMAX) ? murty_ws_particle(nodes) : murty_ws_carve(bigws, n);   // (host: bigws holds murty_big_bytes(n))
	for (int i = lane; i < MURTY_OUT; i += 64) ws.logcomp[i] = 1e300;
	const int stride = (mode == 1) ? 5 : n;
	double* mat = (mode == 1) ? nd.reduced : nd.profit;
	for (int e = lane; e < n * n; e += 64) mat[(e / n) * stride + (e % n)] = matrix[e];
	lds_fence();
	int m;
	if (mode == 1) {
		m = cluster_enumerate_wave(mat, n, modelsize, ws.logcomp, lane, [&](int k, unsigned int perm) {
			if (k < maxcount && lane < n) assignments[k * n + lane] = pk_get(perm, lane);
		});
	}
	else {
		m = wave_murty_any<0>(ws, nd, n, lane, [&](int k, bool unsolved, auto colof) {
			if (k >= maxcount) return;
			for (int r = lane; r < n; r += 64) assignments[k * n + r] = unsolved ? -1 : colof(r);
		});
	}
	for (int k = lane; k < m && k < maxcount; k += 64) values[k] = ws.logcomp[k];
	if (lane == 0) *count = m;
}

// LDS layout of k_weight_alpha, shared with the host so the launch sizes it identically.
// Arrays indexed by landmark live in LDS while the map estimate has at most ALPHA_JL landmarks; a larger
// estimate (up to Jcap) moves them to a per-particle slab in HBM, reached through the same (flat) pointers.
#define ALPHA_JL 256
#ifndef ALPHA_DEFER_ROWS
#define ALPHA_DEFER_ROWS 10   // a particle with an association cluster of more rows than this is left to the big-cluster workers (DEFER, below)
#endif

struct AlphaLds {
	int zs, red, lm, pick, scr;                 // persistent, offsets in doubles
	int p1_keyw, p1_sortw, p1_dw, p1_sortsrc, p1_selidx, ns;   // phase 1 (ints: sortsrc[ns], dsrc[JL]; selidx[ns])
	int p2_tile, p2_part;                       // phase 2
	int p3_zh, p3_pdj, p3_lmd, p3_res, p3_adj, p3_adjT, p3_mem, p3_int, p3_x;   // phase 3 (ints: labl[JL], roots[JL], cnt[JL], labz[MP]); x = mats | murty
	int bytes;
};

__host__ __device__ inline AlphaLds alpha_lds(int MP, int ncap)
{
	const int JL = ALPHA_JL, MW = MP / 64;
	AlphaLds l;
	l.zs   = 0;
	l.red  = l.zs + 3 * MP;
	l.lm   = l.red + 256;              // red[256]: reduction scratch, histogram of the select, exp table of the cluster sums
	l.pick = l.lm + 3 * JL;
	l.scr  = l.pick + JL / 2;
	int ns = 2;
	while (ns < ncap) ns <<= 1;                 // width of the bitonic sort
	l.ns = ns;
	l.p1_keyw = l.scr;
	l.p1_sortw = l.p1_keyw + ncap;
	l.p1_dw = l.p1_sortw + ns;
	l.p1_sortsrc = l.p1_dw + JL;
	l.p1_selidx = l.p1_sortsrc + (ns + JL + 1) / 2;
	int ph1 = l.p1_selidx + ns / 2 - l.scr;
	l.p2_tile = l.scr;
	l.p2_part = l.scr;
	int ph2 = 0;
	l.p3_zh = l.scr;
	l.p3_pdj = l.p3_zh + 3 * JL;
	l.p3_lmd = l.p3_pdj + JL;
	l.p3_res = l.p3_lmd + JL;
	l.p3_adj = l.p3_res + JL;
	l.p3_adjT = l.p3_adj + JL * MW;
	l.p3_mem = l.p3_adjT + MP * (JL / 64);
	l.p3_int = l.p3_mem + 2 * JL;
	l.p3_x   = l.p3_int + (3 * JL + MP + 1) / 2;
	int xs = 25 * 4 > MURTY_LDS_DOUBLES ? 25 * 4 : MURTY_LDS_DOUBLES;
	int ph3 = l.p3_x + xs + 2 - l.scr;
	int mx = ph1 > ph2 ? ph1 : ph2;
	mx = mx > ph3 ? mx : ph3;
	l.bytes = (l.scr + mx) * 8;
	return l;
}

// per-particle HBM slab used instead of LDS when J > ALPHA_JL (doubles; ints packed two per double)
__host__ __device__ inline size_t alpha_jscratch_doubles(int Jcap)
{
	// lm 3J, dw J, zh 3J, lpd J, res J, adj 4J, part 4J  |  pick, dsrc, labl, roots : 4J ints  |  lmd J, memL J, memZ J, cnt J ints
	return (size_t) 23 * Jcap;
}

// QUASI = false: BestMapEstimate + SetLogLikelihood of particle p (k_alpha_assoc).
// QUASI = true : QuasiSetLogLikelihood (PHDNavigator.cs:526-713, value) of candidate pose p against one
//                given landmark set — the same association sum with everything fully visible: constant PD (:574-575),
//                unit-weight measurement Gaussians (:583), detection gate 12 (:615). The map estimate is an input here.
// GRAD (with QUASI): also the pose gradient (:543-548). Every cluster is then enumerated literally and in the
//                reference's order by wave 0 — TemperedAverage rewrites logcomp in place (MatrixExtensions.cs:429-431)
//                and normalises over the whole array, so each cluster sees what the previous ones left.
// gws (gradient mode): QGRAD_LDS_DOUBLES of LDS for the clusters of up to 5 rows — their matrix, Jacobians, member lists
// and per-pairing gradient vectors. (They went through the particle's workspace in HBM, as the large clusters' do: a
// dozen dependent trips to memory per cluster, and a typical pose has forty clusters of two or three rows, replayed in
// order by one wave.)
#define QGRAD_WAVE_DOUBLES (25 + 5 * 18 + 120 * 6 + 120 + 9)   // per wave: matrix, Jacobians, gradient vectors, log components | member lists
#define QGRAD_LDS_DOUBLES (4 * QGRAD_WAVE_DOUBLES)           // (an even count: the dynamic LDS behind it stays 16-byte aligned)
#define QGRAD_G2_CLUSTERS 64                                  // headers / weights of the first pass kept in LDS for the ordered replay
#define QGRAD_G2_WEIGHTS 384
#define QGRAD_HDR 10                                         // doubles per cluster in the particle's scratch: pairings, G[6], list offset, finite
// DEFER (the step's k_alpha_assoc; 0: everything here): the ordered replay of the clusters of more than 5 rows — best-first
//                enumeration by ONE wave, up to 200 assignment problems per cluster — is rare per particle and enormous when
//                it happens (config S: 5 % of the particles, 15 times the median workgroup's lifetime: the launch waited for
//                them). 1: a particle that needs it is put on the launch's list (StepBufs::biglist) and gets no set
//                log-likelihood here; 2: k_alpha_big, one workgroup per listed particle (pin >= 0), runs the body again WITH the
//                replay — on a stream of its own, beside k_alpha_density, which does not need the value (k_alpha_combine does).
#ifndef DENS_JL
#define DENS_JL 128   // landmarks whose partial sums stay in LDS
#endif
// `helper_go` (k_particle_chain with a helper workgroup per particle, a.dsplit): where the map estimate is final the body
// publishes it and — if the particle's helper has reported, from the same XCD — hands the density sums (alpha_density_body) to it,
// which then run BESIDE the association below instead of behind it; *helper_go says whether it did.
__device__ __forceinline__ unsigned int my_xcd() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u; }   // HW_REG_XCC_ID

template <int ZB, bool QUASI, bool GRAD = false, int TAG = 0, int DEFER = 0>
__device__ __forceinline__ void alpha_assoc_body(const DevParams& prm, const StepBufs& a, int ncap, double* smem, double* gws = nullptr, int pin = -1,
                                                 int* helper_go = nullptr)
{
	constexpr int MP = ZB * 64;
	constexpr int MW = ZB;   // 64-bit adjacency words per landmark
	constexpr int JL = ALPHA_JL;
	const AlphaLds lay = alpha_lds(MP, ncap);
	double* zs   = smem + lay.zs;          // [MP][3] measurements
	double* red  = smem + lay.red;         // [256] reduction scratch
	double* etab = red;                    // [256] exp table (filled before the cluster sums, once `red` is idle)
	__shared__ int s_J, s_changed, s_nroots, s_big, s_huge;
	__shared__ double s_ccount, s_total;
	__shared__ int s_ebump, s_g2lds;                                                    // gradient mode (see the first pass below)
	__shared__ double s_g2[GRAD ? QGRAD_G2_CLUSTERS * QGRAD_HDR + QGRAD_G2_WEIGHTS : 2];
	if (GRAD && threadIdx.x == 0) s_g2lds = 0;

	const int p = (pin >= 0) ? pin : a.p0 + blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const int M = a.M, cap = a.cap;
	const MixView vout = bank_view(a, SEL_OUT);
	const Bank bin = bank_of(a, SEL_IN);
	const int no = QUASI ? 0 : vout.count[p];
	bool deferred = false;   // (workgroup-uniform) DEFER == 1: this particle is left to k_alpha_big
	const size_t sbo = (size_t) p * cap;
	const PoseD pose = load_pose(QUASI ? a.qposes + (size_t) p * 7 : bin.poses + (size_t) p * 7);

	for (int k = tid; k < MP * 3; k += 256) zs[k] = (k < M * 3) ? a.z[k] : 0.0;
	// (the helper's word, asked for here and looked at where the map estimate is final: the trip to memory is off the path)
	unsigned int helper_word = 0;
	if (!QUASI && helper_go && tid == 0) helper_word = __hip_atomic_load(a.dsync + 3 * (size_t) p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

	PHD_STAMP_DECL;
	PHD_STAMP(0);
	// ---- phase 1: BestMapEstimate (Map.cs:119-142)
	double* keyw    = smem + lay.p1_keyw;              // [ncap] weights in map order
	double* sortw   = smem + lay.p1_sortw;             // [ncap] weights, stable descending
	int*    sortsrc = (int*) (smem + lay.p1_sortsrc);  // [ncap]
	// block-wide sum in a fixed order (thread partials, then a tree)
	auto block_sum = [&](double v) {
		red[tid] = v;
		__syncthreads();
		for (int s = 128; s > 0; s >>= 1) {
			if (tid < s) red[tid] += red[tid + s];
			__syncthreads();
		}
		double r = red[0];
		__syncthreads();
		return r;
	};
	double wpart = 0;
	for (int c = tid; c < no; c += 256) {
		double wc = a.outw[sbo + c];   // (k_prune_merge's plane of the pruned weights: the records themselves are read for the picks only)
		keyw[c] = wc;
		wpart += wc;
	}
	// ExpectedSize (Map.cs:61-71) and size = (int) ExpectedSize (:126). The reference adds the weights one by
	// one; the tree sum differs from that by a few ulp, which can only change the integer part when the sum
	// sits within that distance of an integer: then, and only then, the weights are re-added in map order.
	{
		double e = block_sum(wpart);
		if (fabs(e - rint(e)) <= 1e-9 * fmax(1.0, fabs(e))) {
			if (tid == 0) {
				double q = 0;
				for (int c = 0; c < no; c++) q += keyw[c];
				s_ccount = q;
			}
			__syncthreads();
			e = s_ccount;
		}
		if (tid == 0) {
			s_ccount = e;
			int J = (int) e;
			if (J < 0) J = 0;
			if (J > a.Jcap) {
				atomicOr(a.flags, PHD_FLAG_J_OVERFLOW);
				J = a.Jcap;
			}
			s_J = QUASI ? min(a.qJ, a.Jcap) : J;
		}
	}
	PHD_STAMP(1);
	__syncthreads();
	const int J = s_J;
	// stable descending order (mlist.Sort, :129). Only the first min(J, no) entries of the sorted list are ever
	// read (the pick below takes at most J of them), so when J <= no the J heaviest are selected (radix select on
	// the weight's bit pattern, ties by map index) and only those are ordered.
	if (J >= 1 && J <= no) {
		// The J heaviest, by ONE histogram over the weights' leading bits (exponent and four mantissa bits from MinWeight up,
		// 500 bins: thirty-one octaves, the last bin takes what lies beyond): the bins above the one the J-th heaviest lies in
		// are taken whole, the entries of that bin are ranked among themselves by (weight, map index) and the first `need` of
		// them taken. (Eight passes of an 8-bit radix select before: three barriers each, 22 k of this kernel's 74 k cycles on
		// config B.)
		constexpr int HB = 500;
		int* const hist   = (int*) red;                                   // [HB + 4] (`red` is idle between the block sums above and the exp table below)
		int* const selidx = (int*) (smem + lay.p1_selidx);                // [J] selected components, unordered
		int* const tl     = sortsrc;                                      // the threshold bin's entries (sortsrc is written only by the ordering below)
		int& s_T = hist[HB];
		int& s_above = hist[HB + 1];
		int& s_nt = hist[HB + 2];
		int& s_ntaken = hist[HB + 3];
		const unsigned int hbase = (unsigned int) (((unsigned long long) __double_as_longlong(prm.minw) << 1) >> 49);
		auto bin_of = [&](double w) {
			const unsigned int k = (unsigned int) (((unsigned long long) __double_as_longlong(w) << 1) >> 49);
			return (int) min(max((int) k - (int) hbase, 0), HB - 1);
		};
		for (int t = tid; t < HB + 4; t += 256) hist[t] = 0;
		__syncthreads();
		for (int c = tid; c < no; c += 256) atomicAdd(&hist[bin_of(keyw[c])], 1);
		__syncthreads();
		if (wv == 0) {   // from the top bin down (lane l: bins 511 - 8 l .. 504 - 8 l): the bin at which J entries are reached
			int cq[8], mine = 0;
#pragma unroll
			for (int q = 0; q < 8; q++) { const int b = 511 - (8 * lane + q); cq[q] = (b < HB) ? hist[b] : 0; mine += cq[q]; }
			int incl = mine;
#pragma unroll
			for (int o = 1; o < 64; o <<= 1) {
				const int y = __shfl_up(incl, o, 64);
				if (lane >= o) incl += y;
			}
			int run = incl - mine;
#pragma unroll
			for (int q = 0; q < 8; q++) {
				if (run < J && J <= run + cq[q]) { s_T = 511 - (8 * lane + q); s_above = run; }
				run += cq[q];
			}
		}
		__syncthreads();
		const int T = s_T, above = s_above, need = J - above;   // `need` of the threshold bin's entries are taken (1 <= need <= its count)
		for (int c = tid; c < no; c += 256) {
			const int b = bin_of(keyw[c]);
			if (b > T) selidx[atomicAdd(&s_ntaken, 1)] = c;            // (the bins above: in any order)
			else if (b == T) tl[atomicAdd(&s_nt, 1)] = c;
		}
		__syncthreads();
		const int nt = s_nt;
		for (int t = tid; t < nt; t += 256) {
			const int ct = tl[t];
			const double wt = keyw[ct];
			int rank = 0;
			for (int u = 0; u < nt; u++) {
				const int cu = tl[u];
				const double wu = keyw[cu];
				rank += (wu > wt || (wu == wt && cu < ct)) ? 1 : 0;
			}
			if (rank < need) selidx[above + rank] = ct;   // (the bins above filled [0, above); ranks are distinct)
		}
		__syncthreads();
		// order the J selected by (weight desc, map index asc): every entry counts the entries before it
		for (int t = tid; t < J; t += 256) {
			const int ct = selidx[t];
			const double wt = keyw[ct];
			int rank = 0;
#pragma unroll 4
			for (int u = 0; u < J; u++) {
				const int cu = selidx[u];
				const double wu = keyw[cu];
				rank += (wu > wt || (wu == wt && cu < ct)) ? 1 : 0;
			}
			sortw[rank]   = wt;
			sortsrc[rank] = ct;
		}
		__syncthreads();
	}
	else if (!QUASI && J > no) {
		// more landmarks than components: the whole list, bitonic sort on the weight's bit pattern, then runs of
		// equal weights put back in map order
		unsigned long long* sv = (unsigned long long*) sortw;   // [NS] sort words, then overwritten by the sorted weights
		int NS = 2;
		while (NS < no) NS <<= 1;
		for (int t = tid; t < NS; t += 256) sv[t] = (t < no) ? prune_pack(keyw[t], t) : 0ull;
		__syncthreads();
		prune_bitonic(sv, NS, tid);
		// runs that agree in the key bits: order by (weight desc, map index asc); the head thread sorts its run
		for (int r = tid; r + 1 < no; r += 256) {
			const unsigned long long kb = prune_kbits(sv[r]);
			if (prune_kbits(sv[r + 1]) != kb || (r > 0 && prune_kbits(sv[r - 1]) == kb)) continue;
			int e = r + 2;
			while (e < no && prune_kbits(sv[e]) == kb) e++;
			for (int x = r + 1; x < e; x++) {
				unsigned long long vx = sv[x];
				int sx = prune_slot(vx);
				double wx = keyw[sx];
				int y = x - 1;
				while (y >= r) {
					int sy = prune_slot(sv[y]);
					if (keyw[sy] > wx || (keyw[sy] == wx && sy < sx)) break;
					sv[y + 1] = sv[y];
					y--;
				}
				sv[y + 1] = vx;
			}
		}
		__syncthreads();
		for (int r = tid; r < no; r += 256) sortsrc[r] = prune_slot(sv[r]);
		__syncthreads();
		for (int r = tid; r < no; r += 256) sortw[r] = keyw[sortsrc[r]];   // same bytes as skey[r]: each thread rewrites its own slots
		__syncthreads();
	}
	PHD_STAMP(2);
	// landmark-indexed arrays: LDS, or the HBM slab of this particle when the estimate is large
	const bool inlds = J <= JL;
	const int  JS = inlds ? JL : a.Jcap;               // stride of the landmark-indexed arrays
	double* gj = a.jscratch + (size_t) p * alpha_jscratch_doubles(a.Jcap);
	double* lm   = inlds ? smem + lay.lm     : gj;                       // [3][JS] landmark means of the map estimate
	double* dw   = inlds ? smem + lay.p1_dw  : gj + 3 * (size_t) JS;     // [JS] derived (w - 1) entries, FIFO
	double* zh   = inlds ? smem + lay.p3_zh  : gj + 4 * (size_t) JS;     // [3][JS] h(m_j)
	double* lpd  = inlds ? smem + lay.p3_pdj : gj + 7 * (size_t) JS;     // [JS] log PD of landmark j
	double* lmd  = inlds ? smem + lay.p3_lmd : gj + 19 * (size_t) JS;    // [JS] log(1 - PD)
	double* res  = inlds ? smem + lay.p3_res : gj + 8 * (size_t) JS;     // [JS] per-cluster log-sum-exp, in cluster order
	unsigned long long* adj = (unsigned long long*) (inlds ? smem + lay.p3_adj : gj + 9 * (size_t) JS);   // [JS][MW]
	int* gi      = (int*) (gj + 17 * (size_t) JS);
	int* pick    = inlds ? (int*) (smem + lay.pick) : gi;                // [JS] component picked for landmark j
	int* dsrc    = inlds ? sortsrc + lay.ns : gi + JS;                   // [JS]
	int* labl    = inlds ? (int*) (smem + lay.p3_int) : gi + 2 * JS;     // [JS]
	int* roots   = inlds ? labl + JL : gi + 3 * JS;                      // [JS]
	int* cnt     = inlds ? labl + 2 * JL : (int*) (gj + 22 * (size_t) JS);   // [JS] members of the cluster rooted at j: landmarks | measurements << 16
	unsigned long long* memL = (unsigned long long*) (inlds ? smem + lay.p3_mem : gj + 20 * (size_t) JS);        // [JS] its first 5 landmarks, 12-bit fields
	unsigned long long* memZ = (unsigned long long*) (inlds ? smem + lay.p3_mem + JL : gj + 21 * (size_t) JS);   // [JS] its first 5 measurements
	int* labz    = (int*) (smem + lay.p3_int) + 3 * JL;                  // [MP]
	double* xreg = smem + lay.p3_x;                                      // mats [25][4]  |  Murty scratch
	// When the largest weight minus one does not exceed the J-th largest weight no appended entry can be
	// picked among the first J: the estimate is simply the J heaviest components, in order.
	const bool straight = J <= no && (J == 0 || !(sortw[0] - 1 > sortw[J - 1]));
	if (QUASI) {}
	else if (straight) {
		for (int j = tid; j < J; j += 256) pick[j] = sortsrc[j];
	}
	else if (tid == 0) {
		// "take the i-th entry, append a copy with w - 1, sort again" (:131-138) is a two-way merge:
		// every appended weight is <= the one it came from, so the appended entries are produced in
		// non-increasing order and form a FIFO merged with the original sorted list; on a tie the
		// original entry stays first (stable sort of an appended element).
		int ia = 0, id = 0, nd = 0;
		for (int j = 0; j < J; j++) {
			bool takeorig;
			if (ia < no && id < nd) takeorig = !(dw[id] > sortw[ia]);
			else takeorig = ia < no;
			double wpick;
			int    src;
			if (takeorig) { wpick = sortw[ia]; src = sortsrc[ia]; ia++; }
			else          { wpick = dw[id];    src = dsrc[id];    id++; }
			dw[nd]   = wpick - 1;
			dsrc[nd] = src;
			nd++;
			pick[j] = src;
		}
	}
	__threadfence_block();
	__syncthreads();
	for (int j = tid; j < J; j += 256) {
		if (QUASI) {
			lm[j] = a.qlm[j * 3]; lm[JS + j] = a.qlm[j * 3 + 1]; lm[2 * JS + j] = a.qlm[j * 3 + 2];
		}
		else {
			int c = pick[j];
			const double2* rc = (const double2*) (vout.rec + (sbo + c) * MIX_REC);   // its mean: the first 32 bytes of the record
			const double2 r0 = rc[0], r1 = rc[1];
			double l0 = r0.y, l1 = r1.x, l2 = r1.y;
			lm[j] = l0; lm[JS + j] = l1; lm[2 * JS + j] = l2;
			double* glm = a.alm + (size_t) p * 3 * a.Jcap;   // for k_alpha_density
			glm[j] = l0; glm[a.Jcap + j] = l1; glm[2 * a.Jcap + j] = l2;
		}
	}
	if (!QUASI && helper_go) {
		// Everything the density sums read is final now: the pruned map, the births, and — just above — the map estimate. The
		// helper gets them through the L2 both workgroups sit behind (so only a helper on this XCD is taken: no write-back of the
		// L2, which cost more than the helper gives): every wave's stores have left the CU (vmcnt) before the barrier below, then
		// thread 0 leaves the word the helper waits for. A map estimate beyond the densities' LDS arrays stays here (its sums would
		// go through the slab this body is still using).
		if (tid == 0) { a.aJ[p] = J; a.account[p] = s_ccount; }
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	}
	__threadfence_block();
	__syncthreads();
	if (!QUASI && helper_go && tid == 0) {
		const int go = (J > 0 && J <= DENS_JL && helper_word == ((a.dstamp << 4) | my_xcd())) ? 1 : 0;
		__hip_atomic_store(a.dsync + 3 * (size_t) p + 1, 2u * a.dstamp + (unsigned int) go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		*helper_go = go;
#ifdef PHD_STAMPS   // (diagnostic build, PHD_STAMP_KERNEL=5: was the helper taken, and when — 100 MHz ticks; see k_particle_chain)
		if (a.stamps && a.stamp_kernel == 5) { a.stamps[(size_t) p * 16 + 12] = go; a.stamps[(size_t) p * 16 + 13] = (double) wall_clock64(); }
#endif
	}

	PHD_STAMP(3);
	// ---- phase 3: SetLogLikelihood (PHDNavigator.cs:462-515) on the matrix of SetLogLikeMatrix (:415-453)
	{
		double* mats = xreg;   // [25][4] one 5x5 matrix per cluster of a tiny map (dead before the Murty path reuses the region)

		constexpr int JW = JL / 64;                                   // words of a transposed adjacency row (LDS case)
		unsigned long long* adjT = (unsigned long long*) (smem + lay.p3_adjT);   // [MP][JW] landmarks gated with measurement k
		for (int j = tid; j < J; j += 256) {
			double m[3] = {lm[j], lm[JS + j], lm[2 * JS + j]}, z[3], l[3];
			measure_perfect(prm, pose, m, z, l);
			zh[j] = z[0]; zh[JS + j] = z[1]; zh[2 * JS + j] = z[2];
			const double pdv = QUASI ? prm.pd : detection_probability_m(prm, z);
			lpd[j] = log(pdv);
			lmd[j] = log(1 - pdv);
#pragma unroll
			for (int b = 0; b < MW; b++) adj[(size_t) j * MW + b] = 0;
			labl[j] = j;
			cnt[j]  = 0;
			memL[j] = 0;
			memZ[j] = 0;
		}
		for (int k = tid; k < M; k += 256) labz[k] = J + k;
		for (int t = tid; t < MP * JW; t += 256) adjT[t] = 0;
		if (tid == 0) { s_nroots = 0; s_big = 0; s_huge = 0; }
		__threadfence_block();
		__syncthreads();
		// detection block: defined iff Mahalanobis(z_k; h(m_j), R) < 5 (:433-442). Measurement per lane, wave w takes the
		// landmarks w, w + 4, ...: a landmark's adjacency words are the ballots of its gate tests (no atomics); only the
		// transposed copy is built from the set bits.
		{
			double kx[ZB], ky[ZB], kz[ZB];
#pragma unroll
			for (int b = 0; b < ZB; b++) {
				const int k = b * 64 + lane;
				kx[b] = zs[k * 3]; ky[b] = zs[k * 3 + 1]; kz[b] = zs[k * 3 + 2];
			}
			const double g2 = QUASI ? prm.g2_quasi : prm.g2_assoc;   // sqrt(q) < 5, :436 (quasi: < 12, :615)
			for (int j = wv; j < J; j += 4) {
				const double h0 = zh[j], h1 = zh[JS + j], h2 = zh[2 * JS + j];
#pragma unroll
				for (int b = 0; b < ZB; b++) {
					const int k = b * 64 + lane;
					const bool gated = k < M && quad_gen(prm.Rinv, h0 - kx[b], h1 - ky[b], h2 - kz[b]) < g2;
					const unsigned long long bal = ballot64(gated);
					if (lane == 0) adj[(size_t) j * MW + b] = bal;
					if (gated && inlds) atomicOr(&adjT[(size_t) k * JW + (j >> 6)], 1ull << (j & 63));
				}
			}
		}
		__threadfence_block();
		__syncthreads();

		PHD_STAMP(4);
		// connected components of the bipartite (landmark, measurement) graph by min-label propagation;
		// a cluster's label ends as its smallest landmark index, which is also its position in the
		// reference's component list (rows with detection entries are inserted first, ascending).
		for (int it = 0; it < J + M + 1; it++) {
			if (tid == 0) s_changed = 0;
			__syncthreads();
			for (int j = tid; j < J; j += 256) {
				int l = labl[j];
#pragma unroll
				for (int b = 0; b < MW; b++) {
					unsigned long long bits = adj[(size_t) j * MW + b];
					while (bits) {
						int k = b * 64 + __ffsll((long long) bits) - 1;
						bits &= bits - 1;
						l = min(l, labz[k]);
					}
				}
				if (l < labl[j]) { labl[j] = l; s_changed = 1; }
			}
			__threadfence_block();
			__syncthreads();
			for (int k = tid; k < M; k += 256) {
				int l = labz[k];
				if (inlds) {
#pragma unroll
					for (int b = 0; b < JW; b++) {
						unsigned long long bits = adjT[(size_t) k * JW + b];
						while (bits) {
							int j = b * 64 + __ffsll((long long) bits) - 1;
							bits &= bits - 1;
							l = min(l, labl[j]);
						}
					}
				}
				else {
					for (int j = 0; j < J; j++) {
						if ((adj[(size_t) j * MW + (k >> 6)] >> (k & 63)) & 1ull) l = min(l, labl[j]);
					}
				}
				if (l < labz[k]) { labz[k] = l; s_changed = 1; }
			}
			__threadfence_block();
			__syncthreads();
			if (!s_changed) break;
			__syncthreads();
		}
		// members of every cluster, appended in arrival order (sorted by the reader)
		for (int j = tid; j < J; j += 256) {
			const int r = labl[j];
			const int slot = atomicAdd(&cnt[r], 1) & 0xffff;
			if (slot < 5) atomicOr(&memL[r], (unsigned long long) j << (12 * slot));
		}
		for (int k = tid; k < M; k += 256) {
			const int r = labz[k];
			if (r < J) {
				const int slot = atomicAdd(&cnt[r], 0x10000) >> 16;
				if (slot < 5) atomicOr(&memZ[r], (unsigned long long) k << (12 * slot));
			}
		}
		if (wv == 0) {   // clusters that hold a detection entry, in ascending order of their first landmark
			int nr = 0;
			for (int j0 = 0; j0 < J; j0 += 64) {
				const int j = j0 + lane;
				bool isroot = false;
				if (j < J) {
					bool has = false;
#pragma unroll
					for (int b = 0; b < MW; b++) has |= adj[(size_t) j * MW + b] != 0;
					isroot = has && labl[j] == j;
				}
				unsigned long long bal = ballot64(isroot);
				if (isroot) roots[nr + __popcll(bal & lanemask_lt())] = j;
				nr += __popcll(bal);
			}
			if (lane == 0) s_nroots = nr;
		}
		exp_tab_init(etab, tid);
		__syncthreads();
		const int nroots = s_nroots;
		const double logmult = prm.logRmult;

		PHD_STAMP(5);
		// clusters with n <= 5 rows: every pairing (PHDNavigator.cs:492-494)
		if (GRAD) {
			if (tid == 0) s_big = 1;   // all clusters go through the ordered replay below
		}
		else if (J >= 5) {
			// `modelsize` = J >= n, so LexicographicalPairing walks all n! pairings once and their log-sum-exp is
			// the log of the permanent of exp(matrix). One thread per cluster; the 5 x 5 matrix sits in registers
			// with every index static: landmark x -> row x and misdetection column x, measurement y -> clutter
			// row 4 - y and column 4 - y (n = nl + nz <= 5 keeps the two ranges apart; a free row keeps a 1 on the
			// diagonal). Landmark rows are scaled by their largest entry.
			for (int ri = tid; ri < nroots; ri += 256) {
				const int root = roots[ri];
				const int cn = cnt[root];
				const int nl = cn & 0xffff, nz = cn >> 16;
				if (nl + nz > 5) {
					res[ri] = NAN;   // solved below by the Murty path
					s_big = 1;
					if (nl + nz > ALPHA_DEFER_ROWS) s_huge = 1;
					continue;
				}
				const unsigned long long Lp = pk_sort5(memL[root], nl), Zp = pk_sort5(memZ[root], nz);   // members, ascending
				double E[25];
#pragma unroll
				for (int e = 0; e < 25; e++) E[e] = (e % 6 == 0) ? 1.0 : 0.0;
				double rsum = 0;
				bool dead = false;
#pragma unroll
				for (int x = 0; x < 4; x++) {
					if (x < nl) {
						const int j = (int) ((Lp >> (12 * x)) & 4095);
						const double md = lmd[j], lp = lpd[j];   // :445, :439
						const double h0 = zh[j], h1 = zh[JS + j], h2 = zh[2 * JS + j];
						double D[4], rx = md;
#pragma unroll
						for (int y = 0; y < 4 - x; y++) {
							D[y] = -INFINITY;
							if (y < nz) {
								const int k = (int) ((Zp >> (12 * y)) & 4095);
								if ((adj[(size_t) j * MW + (k >> 6)] >> (k & 63)) & 1ull) {
									double dist = sqrt(quad_gen(prm.Rinv, h0 - zs[k * 3], h1 - zs[k * 3 + 1], h2 - zs[k * 3 + 2]));
									D[y] = lp + logmult - 0.5 * dist * dist;
									rx = fmax(rx, D[y]);
								}
							}
						}
						if (rx == -INFINITY) dead = true;   // a row with no finite entry: every pairing is -inf
						else {
							rsum += rx;
							E[x * 6] = exp_neg(md - rx, etab);
#pragma unroll
							for (int y = 0; y < 4 - x; y++) {
								if (y < nz) E[x * 5 + 4 - y] = (D[y] == -INFINITY) ? 0.0 : exp_neg(D[y] - rx, etab);
							}
						}
					}
				}
#pragma unroll
				for (int y = 0; y < 4; y++) {
					if (y < nz) {
						E[(4 - y) * 6] = prm.kappa;   // :449
#pragma unroll
						for (int x = 0; x < 4 - y; x++) {
							if (x < nl) E[(4 - y) * 5 + x] = 1.0;   // zero quadrant, :480-488
						}
					}
				}
				// permanent by rows over the sets of used columns
				double f[32];
#pragma unroll
				for (int m = 0; m < 32; m++) f[m] = 0;
				f[0] = 1;
#pragma unroll
				for (int i = 0; i < 5; i++) {
#pragma unroll
					for (int m = 0; m < 32; m++) {
						if (__builtin_popcount(m) == i) {
#pragma unroll
							for (int c = 0; c < 5; c++) {
								if (!((m >> c) & 1)) f[m | (1 << c)] = fma(f[m], E[i * 5 + c], f[m | (1 << c)]);
							}
						}
					}
				}
				res[ri] = dead ? -INFINITY : rsum + log(f[31]);
			}
		}
		else if (wv == 0) {
			// a map estimate of fewer than 5 landmarks: `modelsize` may cut the enumeration short
			// (GraphCombinatorics.cs:293-299), so these few clusters are enumerated literally, one lane each
			for (int r0 = 0; r0 < nroots; r0 += 64) {
				int ri = r0 + lane;
				if (ri < nroots) {
					int root = roots[ri];
					unsigned long long Lp = 0, Zp = 0;
					int nl = 0, nz = 0, nrow = 0;
					for (int j = root; j < J; j++) {
						if (labl[j] == root) { if (nl < 5) Lp |= (unsigned long long) j << (12 * nl); nl++; }
					}
					for (int k = 0; k < M; k++) {
						if (labz[k] == root) { if (nz < 5) Zp |= (unsigned long long) k << (12 * nz); nz++; }
					}
					nrow = nl + nz;
					if (nrow > 5) {
						res[ri] = NAN;   // solved below by the Murty path
						s_big = 1;
						if (nrow > ALPHA_DEFER_ROWS) s_huge = 1;
					}
					else {
						double* mat = mats + lane;   // entry e at mat[e * 4] (fewer than 5 landmarks: at most 4 clusters)
						for (int e = 0; e < 25; e++) mat[e * 4] = -INFINITY;
						for (int x = 0; x < nl; x++) {
							int j = (int) ((Lp >> (12 * x)) & 4095);
							for (int y = 0; y < nz; y++) {
								int k = (int) ((Zp >> (12 * y)) & 4095);
								if ((adj[(size_t) j * MW + (k >> 6)] >> (k & 63)) & 1ull) {
									double dist = sqrt(quad_gen(prm.Rinv, zh[j] - zs[k * 3], zh[JS + j] - zs[k * 3 + 1],
									                            zh[2 * JS + j] - zs[k * 3 + 2]));
									mat[(x * 5 + y) * 4] = lpd[j] + logmult - 0.5 * dist * dist;   // :439
								}
							}
							mat[(x * 5 + nz + x) * 4] = lmd[j];   // :445
						}
						for (int y = 0; y < nz; y++) {
							mat[((nl + y) * 5 + y) * 4] = prm.logkappa;   // :449
							for (int x = 0; x < nl; x++) mat[((nl + y) * 5 + nz + x) * 4] = 0;   // :480-488
						}
						res[ri] = cluster_enumerate(mat, 4, nrow, J);
					}
				}
			}
		}
		__syncthreads();
		if (GRAD && gws && inlds) {
			// Gradient mode, first pass: what a cluster of up to 5 rows contributes does not depend on the clusters before it
			// except through the normalisation of TemperedAverage (MatrixExtensions.cs:400-440, over the whole logcomp array):
			// its pairings, their log components and gradient vectors, the log-sum-exp, the weights exp(l - max) and the
			// weighted gradient sum G are computed here, the clusters shared among the four waves; the ordered replay below
			// (one wave) then only moves the weights into logcomp and divides G by the norm. Larger clusters (MurtyPairing with
			// its cut on the array's stale entries) are left to the replay whole.
			double* const wsm = gws + wv * QGRAD_WAVE_DOUBLES;
			double* const gmat = wsm;                                  // [5][5]
			double* const gjp = wsm + 25;                              // [5][18]
			double (*gdv)[6] = (double (*)[6]) (wsm + 25 + 5 * 18);    // [120][6]
			double* const glc = wsm + 25 + 5 * 18 + 120 * 6;           // [120]
			int* const gL = (int*) (glc + 120);
			int* const gZ = gL + 8;
			// the particle's scratch: cluster headers | MeasurementJacobianP of every landmark (:591), computed here by all
			// threads at once rather than by the one or two lanes a cluster has landmarks for | the clusters' weight lists
			double* const jpall = gj + (size_t) QGRAD_HDR * JL;
			double* const elist = jpall + (size_t) 18 * JL;
			const int ecap = (int) alpha_jscratch_doubles(a.Jcap) - (QGRAD_HDR + 18) * JL;
			const int gt = (lane < 6) ? lane : 0;
			if (tid == 0) s_ebump = 0;
			if (ecap > 0) {
				for (int j = tid; j < J; j += 256) {
					const double m3[3] = {lm[j], lm[JS + j], lm[2 * JS + j]};
					jacobian_p(prm, pose, m3, jpall + (size_t) j * 18);
				}
			}
			__threadfence_block();
			__syncthreads();
			// A cluster of ONE landmark and ONE measurement — in a scene whose landmarks stand apart, nearly every cluster — needs
			// no wave: its two pairings are (detection | -) and (misdetection | clutter), in this order when the map estimate
			// has at least as many landmarks as the cluster has rows (LexicographicalPairing's `modelsize`; J >= 5 as in the
			// value kernel's register path). One LANE per such cluster computes what the loop below computes for it, in the same
			// arithmetic and order; the loop then passes it by.
			const bool pairfast = J >= 5 && ecap > 0;
			if (pairfast) {
				for (int ri = tid; ri < nroots; ri += 256) {
					const int root = roots[ri];
					if (cnt[root] != 0x10001) continue;
					const int j = (int) (memL[root] & 4095), k = (int) (memZ[root] & 4095);
					double* const h = gj + (size_t) ri * QGRAD_HDR;
					const bool gated = (adj[(size_t) j * MW + (k >> 6)] >> (k & 63)) & 1ull;
					double c0 = -INFINITY;
					if (gated) {
						double dist = sqrt(quad_gen(prm.Rinv, zh[j] - zs[k * 3], zh[JS + j] - zs[k * 3 + 1], zh[2 * JS + j] - zs[k * 3 + 2]));
						c0 = lpd[j] + logmult - 0.5 * dist * dist;
					}
					c0 = (0.0 + c0) + 0.0;                            // AssignmentValue: rows in order, from 0
					const double c1 = (0.0 + lmd[j]) + prm.logkappa;
					// LogSumExp(logcomp, 0, 2), MatrixExtensions.cs:361-389
					const double mx = fmax(fmax(-INFINITY, c0), c1);
					const bool finite = !(isinf(mx) && mx < 0);
					double lse = -INFINITY, w0 = c0, w1 = c1;
					if (finite) {
						double value = 0;
						value += exp(c0 - mx);
						value += exp(c1 - mx);
						lse = mx + log(value);
						w0 = exp_neg(c0 - mx, etab);
						w1 = exp_neg(c1 - mx, etab);
					}
					res[ri] = lse;
					const int off = atomicAdd(&s_ebump, 2);
					if (off + 2 > ecap) { h[0] = -2; continue; }     // no room for its weights: the replay enumerates it again
					elist[off] = w0;
					elist[off + 1] = w1;
					double G[6] = {0, 0, 0, 0, 0, 0};
					if (finite && gated) {
						const double nu[3] = {zs[k * 3] - zh[j], zs[k * 3 + 1] - zh[JS + j], zs[k * 3 + 2] - zh[2 * JS + j]};
						const double* Jp = jpall + (size_t) j * 18;
#pragma unroll
						for (int t = 0; t < 6; t++) G[t] = w0 * (0.0 + pair_gradient(prm, nu, Jp, t));
					}
					h[0] = 2.0;
#pragma unroll
					for (int t = 0; t < 6; t++) h[1 + t] = G[t];
					h[7] = (double) off;
					h[8] = finite ? 1.0 : 0.0;
				}
			}
			PHD_STAMP(9);
			for (int ri = wv; ri < nroots; ri += 4) {
				const int root = roots[ri];
				if (pairfast && cnt[root] == 0x10001) continue;
				double* const h = gj + (size_t) ri * QGRAD_HDR;
				int nl = 0, nz = 0;
				for (int j0 = root; j0 < J; j0 += 64) {
					const int j = j0 + lane;
					const bool in = j < J && labl[j] == root;
					const unsigned long long bal = ballot64(in);
					const int pos = nl + __popcll(bal & lanemask_lt());
					if (in && pos < 6) gL[pos] = j;
					nl += __popcll(bal);
				}
				for (int k0 = 0; k0 < M; k0 += 64) {
					const int k = k0 + lane;
					const bool in = k < M && labz[k] == root;
					const unsigned long long bal = ballot64(in);
					const int pos = nz + __popcll(bal & lanemask_lt());
					if (in && pos < 6) gZ[pos] = k;
					nz += __popcll(bal);
				}
				const int nrow = nl + nz;
				if (nrow > 5) {
					if (lane == 0) h[0] = -1;   // the replay takes it whole
					continue;
				}
				lds_fence();
				if (lane < nrow * nrow) {
					const int x = lane / nrow, y = lane - x * nrow;
					double v = -INFINITY;
					if (x < nl) {
						const int j = gL[x];
						if (y < nz) {
							const int k = gZ[y];
							if ((adj[(size_t) j * MW + (k >> 6)] >> (k & 63)) & 1ull) {
								double dist = sqrt(quad_gen(prm.Rinv, zh[j] - zs[k * 3], zh[JS + j] - zs[k * 3 + 1], zh[2 * JS + j] - zs[k * 3 + 2]));
								v = lpd[j] + logmult - 0.5 * dist * dist;
							}
						}
						else if (y - nz == x) v = lmd[j];
					}
					else {
						if (y < nz) { if (y == x - nl) v = prm.logkappa; }
						else v = 0;
					}
					gmat[x * 5 + y] = v;
				}
				if (ecap <= 0) {
					if (lane == 0) h[0] = -2;   // (a scratch too small for the lists: everything goes to the replay)
					continue;
				}
				int mcount;
				double mx = -INFINITY, value = 0;
				if (nrow <= J) {
					// The map estimate has at least as many landmarks as the cluster has rows: LexicographicalPairing walks all n!
					// pairings in plain lexicographic order from the identity. One LANE per pairing instead of the walk: first the
					// pair gradients dlldp[x, y] (:605-608), once each, into the Jacobians' place (lane 6 (x nz + y) + t: component
					// t); then every lane unranks its pairing and adds up its log component and its gradient vector — rows in
					// order, as the walk does.
					double* const pgt = gjp;   // [x * 4 + y][6]; x * 4 + y <= 12 with nl + nz <= 5
					const int npair = nl * nz;
					bool g = false;
					{
						const int pr = lane / 6, t = lane - 6 * pr;
						if (pr < npair) {
							const int x = pr / nz, y = pr - x * nz;
							const int j = gL[x], k = gZ[y];
							g = (adj[(size_t) j * MW + (k >> 6)] >> (k & 63)) & 1ull;
							double v = 0;
							if (g) {
								const double nu[3] = {zs[k * 3] - zh[j], zs[k * 3 + 1] - zh[JS + j], zs[k * 3 + 2] - zh[2 * JS + j]};
								v = pair_gradient(prm, nu, jpall + (size_t) j * 18, t);
							}
							pgt[(x * 4 + y) * 6 + t] = v;
						}
					}
					const unsigned long long gb = ballot64(g);
					unsigned int gmask = 0;   // bit x * 4 + y: the pair is gated
					for (int pr = 0; pr < npair; pr++) {
						if ((gb >> (6 * pr)) & 1ull) gmask |= 1u << ((pr / nz) * 4 + (pr % nz));
					}
					lds_fence();
					mcount = 1;
					for (int i = 2; i <= nrow; i++) mcount *= i;
					double vv[2] = {-INFINITY, -INFINITY};
#pragma unroll
					for (int pass = 0; pass < 2; pass++) {
						const int m = lane + 64 * pass;
						if (m < mcount) {
							const unsigned int perm = pk_unrank(m, nrow);
							double v = 0;
							for (int i = 0; i < nrow; i++) v += gmat[i * 5 + pk_get(perm, i)];   // AssignmentValue
							double acc[6] = {0, 0, 0, 0, 0, 0};
							for (int x = 0; x < nl; x++) {
								const int y = pk_get(perm, x);
								if (y < nz && ((gmask >> (x * 4 + y)) & 1u)) {
#pragma unroll
									for (int t = 0; t < 6; t++) acc[t] += pgt[(x * 4 + y) * 6 + t];
								}
							}
							glc[m] = v;
#pragma unroll
							for (int t = 0; t < 6; t++) gdv[m][t] = acc[t];
							vv[pass] = v;
						}
					}
					// LogSumExp(logcomp, 0, m), MatrixExtensions.cs:361-389: the terms by their lanes, the sum in the walk's order
					mx = fmax(vv[0], vv[1]);
#pragma unroll
					for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
					if (!(isinf(mx) && mx < 0)) {
						const double e0 = (lane < mcount) ? exp(vv[0] - mx) : 0.0, e1 = (lane + 64 < mcount) ? exp(vv[1] - mx) : 0.0;
						auto rld = [](double v, int l) {
							return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
						};
						for (int i = 0; i < mcount; i++) value += (i < 64) ? rld(e0, i) : rld(e1, i - 64);
					}
					lds_fence();
				}
				else {
					for (int e = lane; e < nl * 18; e += 64) gjp[e] = jpall[(size_t) gL[e / 18] * 18 + e % 18];
					lds_fence();
					mcount = cluster_enumerate_wave(gmat, nrow, J, glc, lane, [&](int m, unsigned int perm) {
						double acc = 0;
						for (int x = 0; x < nl; x++) {
							const int y = pk_get(perm, x);
							if (y >= nz) continue;
							const int j = gL[x], k = gZ[y];
							if (!((adj[(size_t) j * MW + (k >> 6)] >> (k & 63)) & 1ull)) continue;
							const double nu[3] = {zs[k * 3] - zh[j], zs[k * 3 + 1] - zh[JS + j], zs[k * 3 + 2] - zh[2 * JS + j]};
							acc += pair_gradient(prm, nu, gjp + x * 18, gt);
						}
						if (lane < 6) gdv[m][lane] = acc;
					});
					// LogSumExp(logcomp, 0, m)
					for (int i = 0; i < mcount; i++) mx = fmax(mx, glc[i]);
					if (!(isinf(mx) && mx < 0)) {
						for (int i = 0; i < mcount; i++) value += exp(glc[i] - mx);   // (as cluster_enumerate does: the value of this mode equals the value kernel's bit for bit)
					}
				}
				const bool finite = !(isinf(mx) && mx < 0);
				const double lse = finite ? mx + log(value) : -INFINITY;
				int off = 0;
				if (lane == 0) off = atomicAdd(&s_ebump, mcount);
				off = __shfl(off, 0, 64);
				if (lane == 0) res[ri] = lse;
				if (off + mcount > ecap) {
					if (lane == 0) h[0] = -2;   // no room for its weights: the replay enumerates it again
					continue;
				}
				lds_fence();
				if (finite) {
					for (int i = lane; i < mcount; i += 64) glc[i] = exp_neg(glc[i] - mx, etab);
					lds_fence();
				}
				for (int i = lane; i < mcount; i += 64) elist[off + i] = glc[i];   // (all -inf: the raw components stay, as in the reference)
				double G = 0;
				if (finite) for (int i = 0; i < mcount; i++) G += glc[i] * gdv[i][gt];
				if (lane < 6) h[1 + lane] = G;
				if (lane == 0) { h[0] = (double) mcount; h[7] = (double) off; h[8] = finite ? 1.0 : 0.0; }
			}
			__threadfence_block();
			__syncthreads();
			PHD_STAMP(10);
			// what the replay reads per cluster, brought into LDS in one go when it is little (it usually is: forty clusters, a
			// few weights each): the replay is one wave walking the clusters in order, every read a dependent trip otherwise
			// (the bump counter also counts the clusters that found no room, off + mcount > ecap: only a list that fits the
			// particle's scratch as a whole is copied — nothing is read behind its end)
			if (nroots <= QGRAD_G2_CLUSTERS && s_ebump <= min(QGRAD_G2_WEIGHTS, ecap)) {
				for (int i = tid; i < nroots * QGRAD_HDR; i += 256) s_g2[i] = gj[i];
				for (int i = tid; i < s_ebump; i += 256) s_g2[QGRAD_G2_CLUSTERS * QGRAD_HDR + i] = elist[i];
				if (tid == 0) s_g2lds = 1;
			}
			__syncthreads();
		}
		PHD_STAMP(8);
		if (DEFER == 1 && s_huge) {
			deferred = true;
			if (tid == 0) a.biglist[1 + atomicAdd(a.biglist, 1)] = p;   // ([0]: entries; emptied by k_normalise_resample)
		}
		else if ((DEFER != 1 || ALPHA_DEFER_ROWS > 5) && s_big) {   // (ALPHA_DEFER_ROWS 5: the main kernel defers every such particle and carries no solver at all)
			// Some cluster has more than 5 rows: it is enumerated best-first (MurtyPairing) under the
			// early-exit test of PHDNavigator.cs:503, which reads logcomp[m] as left behind by the clusters
			// before it. So wave 0 replays the clusters in order up to the last such cluster, keeping the
			// shared logcomp array; the other waves wait.
			if (wv == 0) {
				const MurtyLds ws = murty_lds(xreg);
				for (int i = lane; i < MURTY_OUT; i += 64) ws.logcomp[i] = 0;   // new double[200], :469
				lds_fence();
				int lastbig = GRAD ? nroots - 1 : -1;
				for (int r = 0; r < nroots && !GRAD; r++) {
					if (isnan(res[r])) lastbig = r;
				}
				MurtyNodes* nodes = a.murty + p;
				const int gt = (lane < 6) ? lane : 0;   // gradient component of this lane
				double gacc = 0;
				auto replay_one = [&](int ri) {
					const int root = roots[ri];
					int nl = 0, nz = 0;
					for (int j0 = root; j0 < J; j0 += 64) nl += __popcll(ballot64(j0 + lane < J && labl[j0 + lane] == root));
					for (int k0 = 0; k0 < M; k0 += 64) nz += __popcll(ballot64(k0 + lane < M && labz[k0 + lane] == root));
					const int nrow = nl + nz;
					// workspace: the particle's own for clusters of up to MURTY_NMAX rows, a block of the association slab beyond
					MurtyWs nd = murty_ws_particle(nodes);
					if (nrow > MURTY_NMAX) {
						const unsigned long long need = murty_big_bytes(nrow);
						unsigned long long off = 0;
						if (lane == 0 && a.bigws && nrow <= MURTY_NBIG) off = atomicAdd(a.bigws_used, need);
						off = __shfl(off, 0, 64);
						if (!a.bigws || nrow > MURTY_NBIG || off + need > a.bigws_bytes) {
							// more rows than the solver takes, or the slab is used up: the step is dropped (PHD_ERR_ASSOCIATION)
							if (lane == 0) { atomicOr(a.flags, PHD_FLAG_BIG_CLUSTER); res[ri] = 0; }
							return;
						}
						nd = murty_ws_carve(a.bigws + off, nrow);
					}
					double (*dvec)[6] = nodes->dvec;
					if (GRAD && gws && nrow <= 5) {   // a small cluster of the gradient replay: everything in LDS
						nd.reduced = gws; nd.jp = gws + 25; dvec = (double (*)[6]) (gws + 25 + 5 * 18);
						nd.L = (int*) (gws + 25 + 5 * 18 + 120 * 6 + 120); nd.Z = nd.L + 8;
					}
					nl = 0; nz = 0;
					for (int j0 = root; j0 < J; j0 += 64) {
						int j = j0 + lane;
						bool in = j < J && labl[j] == root;
						unsigned long long bal = ballot64(in);
						int pos = nl + __popcll(bal & lanemask_lt());
						if (in) nd.L[pos] = j;
						nl += __popcll(bal);
					}
					for (int k0 = 0; k0 < M; k0 += 64) {
						int k = k0 + lane;
						bool in = k < M && labz[k] == root;
						unsigned long long bal = ballot64(in);
						int pos = nz + __popcll(bal & lanemask_lt());
						if (in) nd.Z[pos] = k;
						nz += __popcll(bal);
					}
					__threadfence_block();
					// the cluster's square matrix: rows = landmarks then clutter rows, columns = measurements
					// then misdetection columns (Compact, SparseMatrix.cs:592-628; zero quadrant :480-488)
					const int stride = (nrow <= 5) ? 5 : nrow;
					double* mat = (nrow <= 5) ? nd.reduced : nd.profit;
					for (int e = lane; e < nrow * nrow; e += 64) {
						int x = e / nrow, y = e - x * nrow;
						double v = -INFINITY;
						if (x < nl) {
							int j = nd.L[x];
							if (y < nz) {
								int k = nd.Z[y];
								if ((adj[(size_t) j * MW + (k >> 6)] >> (k & 63)) & 1ull) {
									double dist = sqrt(quad_gen(prm.Rinv, zh[j] - zs[k * 3], zh[JS + j] - zs[k * 3 + 1],
									                            zh[2 * JS + j] - zs[k * 3 + 2]));
									v = lpd[j] + logmult - 0.5 * dist * dist;
								}
							}
							else if (y - nz == x) v = lmd[j];
						}
						else {
							if (y < nz) { if (y == x - nl) v = prm.logkappa; }
							else v = 0;
						}
						mat[x * stride + y] = v;
					}
					__threadfence_block();
					// gradient mode: dlogcompdp[m] = sum over the rows of dcomp[row, pairing[row]] (:676-678); lane t keeps
					// component t. Only landmark rows paired with a gated measurement hold a vector (the rest is the default 0).
					if (GRAD) {
						for (int x = lane; x < nl; x += 64) {
							const int j = nd.L[x];
							const double m3[3] = {lm[j], lm[JS + j], lm[2 * JS + j]};
							jacobian_p(prm, pose, m3, nd.jp + x * 18);   // :591
						}
						__threadfence_block();
					}
					auto pairing_gradient = [&](auto colof) {
						double acc = 0;
						for (int x = 0; x < nl; x++) {
							const int y = colof(x);
							if (y >= nz) continue;
							const int j = nd.L[x], k = nd.Z[y];
							if (!((adj[(size_t) j * MW + (k >> 6)] >> (k & 63)) & 1ull)) continue;
							const double nu[3] = {zs[k * 3] - zh[j], zs[k * 3 + 1] - zh[JS + j], zs[k * 3 + 2] - zh[2 * JS + j]};
							acc += pair_gradient(prm, nu, nd.jp + x * 18, gt);
						}
						return acc;
					};
					int mcount = -1;
					if (nrow <= 5) {
						if (GRAD) {
							mcount = cluster_enumerate_wave(mat, nrow, J, ws.logcomp, lane, [&](int m, unsigned int perm) {
								const double g = pairing_gradient([&](int x) { return pk_get(perm, x); });
								if (lane < 6) dvec[m][lane] = g;
							});
						}
						else {
							// only its logcomp entries matter here (res[ri] is already known)
							if (lane == 0) cluster_enumerate(mat, 1, nrow, J, ws.logcomp);
							lds_fence();
						}
					}
					else {
						auto hook = [&](int m, bool unsolved, auto colof) {
							if (!GRAD) return;
							double g = 0;
							if (!unsolved) g = pairing_gradient(colof);
							if (lane < 6) dvec[m][lane] = g;
						};
						// (the step's main kernel replays clusters of at most ALPHA_DEFER_ROWS rows only: the one-row-per-lane
						// solver, none of the frames of the two- and four-row ones)
						if (DEFER == 1) mcount = wave_murty<1, TAG>(ws, nd, nrow, lane, hook);
						else mcount = wave_murty_any<TAG>(ws, nd, nrow, lane, hook);
					}
					if (mcount >= 0) {
						// LogSumExp(logcomp, 0, m), MatrixExtensions.cs:361-389
						double mx = -INFINITY, value = 0;
						for (int i = 0; i < mcount; i++) mx = fmax(mx, ws.logcomp[i]);
						double lse;
						if (isinf(mx) && mx < 0) lse = -INFINITY;
						else {
							for (int i = 0; i < mcount; i++) value += exp(ws.logcomp[i] - mx);
							lse = mx + log(value);
						}
						if (lane == 0) res[ri] = lse;
						lds_fence();
						if (GRAD && !(isinf(mx) && mx < 0)) {
							// TemperedAverage(dlogcompdp, logcomp, 0, m) (:707; MatrixExtensions.cs:400-440): logcomp[0, m) becomes
							// exp(l - max) for good; a.qavg 0: divided by the Euclidean norm of all 200 entries (Accord's vector
							// Normalize, as the source reads), 1: by their sum over [0, m)
							for (int i = lane; i < mcount; i += 64) ws.logcomp[i] = exp(ws.logcomp[i] - mx);
							lds_fence();
							double part = 0;
							if (a.qavg == 0) {
								for (int i = lane; i < MURTY_OUT; i += 64) part += ws.logcomp[i] * ws.logcomp[i];
							}
							else {
								for (int i = lane; i < mcount; i += 64) part += ws.logcomp[i];
							}
#pragma unroll
							for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
							const double norm = (a.qavg == 0) ? sqrt(part) : part;
							double val = 0;
							for (int i = 0; i < mcount; i++) {
								const double wn = (norm == 0) ? ws.logcomp[i] : ws.logcomp[i] / norm;
								val += wn * dvec[i][gt];
							}
							gacc += val;
						}
					}
				};
				if (GRAD && gws && inlds) {
					// The clusters the four waves have already enumerated: only what depends on the clusters before them is left —
					// their weights take their place in logcomp, which TemperedAverage normalises as a whole (the stale entries of
					// earlier, longer clusters included). Here the array lives in registers (entry lane + 64 q): a cluster costs a
					// predicated copy and the lane's share of the norm, written to a row of LDS; the rows of a run of clusters are
					// then summed by one lane each and the clusters' G / norm added in cluster order. A cluster left to the replay
					// whole (more than 5 rows, or no room for its weights) ends the run: the registers go to the LDS array it works
					// on and come back.
					constexpr int LQ = (MURTY_OUT + 63) / 64, SEG = 32, ROW = 65;
					double* const prow = gws + QGRAD_WAVE_DOUBLES;   // [SEG][ROW] (the other waves' workspaces of the first pass: free now)
					static_assert(SEG * ROW <= 3 * QGRAD_WAVE_DOUBLES, "the rows of a run do not fit the three idle workspaces");
					double l[LQ];
#pragma unroll
					for (int q = 0; q < LQ; q++) l[q] = 0;
					int ri = 0, maxm = 0;   // (entries from maxm on are still the zeros the array started with)
					double T = 0;           // TemperedAverage's first reading (a.qavg 0): this lane's share of the squares of the entries from 64 on
					// one run of clusters, from r0 on; returns their number. (Inlined once per place the headers and weight lists
					// can be in — LDS or the particle's scratch —, so that each copy knows its address space: through one pointer
					// chosen at run time every access was a flat one.)
					auto light_run = [&](const double* const hb, const double* const eb, const int r0) __attribute__((always_inline)) {
						// the headers of the run: lane s reads cluster r0 + s; the run ends before the first cluster left to the replay
						const int avail = min(SEG, lastbig + 1 - r0);
						int hm = 0, hoff = 0;
						if (lane < avail) {
							const double* h = hb + (size_t) (r0 + lane) * QGRAD_HDR;
							hm = (int) h[0];
							hoff = (int) h[7];
						}
						const unsigned long long whole = ballot64(lane < avail && hm < 0);
						const int ns = whole ? min(avail, __ffsll((long long) whole) - 1) : avail;
						for (int s0 = 0; s0 < ns; s0 += 4) {   // four clusters at a time: their weights are fetched before any row is stored
							int mm[4], mx4 = 0;
#pragma unroll
							for (int q = 0; q < 4; q++) {
								mm[q] = (s0 + q < ns) ? __builtin_amdgcn_readlane(hm, min(s0 + q, ns - 1)) : 0;
								mx4 = max(mx4, mm[q]);
							}
							if (mx4 <= 64) {
								// (the usual case — at most 64 pairings each: one register per lane; the entries from 64 on stay, T is their share)
								double ev[4];
#pragma unroll
								for (int q = 0; q < 4; q++) {
									const double* el = eb + __builtin_amdgcn_readlane(hoff, min(s0 + q, ns - 1));
									ev[q] = (lane < mm[q]) ? el[lane] : 0.0;
								}
#pragma unroll
								for (int q = 0; q < 4; q++) {
									if (s0 + q < ns) {
										if (lane < mm[q]) l[0] = ev[q];
										prow[(s0 + q) * ROW + lane] = (a.qavg == 0) ? fma(l[0], l[0], T) : ((lane < mm[q]) ? l[0] : 0.0);
									}
								}
							}
							else {
#pragma unroll
								for (int q = 0; q < 4; q++) {
									if (s0 + q < ns) {
										const int m = mm[q];
										const double* el = eb + __builtin_amdgcn_readlane(hoff, min(s0 + q, ns - 1));
										double sum = 0;
										T = 0;
#pragma unroll
										for (int qq = 0; qq < LQ; qq++) {
											const int i = lane + 64 * qq;
											if (i < m) { l[qq] = el[i]; sum += l[qq]; }
											if (qq > 0) T = fma(l[qq], l[qq], T);
										}
										prow[(s0 + q) * ROW + lane] = (a.qavg == 0) ? fma(l[0], l[0], T) : sum;
									}
								}
							}
							maxm = max(maxm, mx4);
						}
						lds_fence();
						if (ns > 0) {
							// the rows summed by one lane each (lanes from maxm on wrote exact zeros when no entry beyond 64 is alive: adding
							// them changes nothing); that lane divides its cluster's G by the norm
							double* const cbuf = prow;   // [SEG][6] once the rows are summed
							bool on = false;
							double c6[6];
							if (lane < ns) {
								const int tn = (maxm <= 64) ? maxm : 64;
								double tot = 0;
#pragma unroll 8
								for (int t = 0; t < tn; t++) tot += prow[lane * ROW + t];
								const double norm = (a.qavg == 0) ? sqrt(tot) : tot;
								const double* h = hb + (size_t) (r0 + lane) * QGRAD_HDR;
#pragma unroll
								for (int t = 0; t < 6; t++) c6[t] = (norm == 0) ? h[1 + t] : h[1 + t] / norm;
								on = h[8] != 0;
							}
							const unsigned long long onb = ballot64(on);
							lds_fence();   // (every row has been read)
							if (lane < ns) {
#pragma unroll
								for (int t = 0; t < 6; t++) cbuf[lane * 6 + t] = c6[t];
							}
							lds_fence();
							// ... and the quotients are added in cluster order (a cluster whose components were all -inf adds nothing)
							for (int s0 = 0; s0 < ns; s0 += 8) {
								double vq[8];
#pragma unroll
								for (int q = 0; q < 8; q++) {
									const double v = cbuf[min(s0 + q, ns - 1) * 6 + gt];
									vq[q] = (s0 + q < ns && ((onb >> (s0 + q)) & 1ull)) ? v : 0.0;
								}
#pragma unroll
								for (int q = 0; q < 8; q++) gacc += vq[q];
							}
						}
						return ns;
					};
					while (ri <= lastbig) {
						const int ns = s_g2lds ? light_run(s_g2, s_g2 + QGRAD_G2_CLUSTERS * QGRAD_HDR, ri)
						                       : light_run(gj, gj + (size_t) (QGRAD_HDR + 18) * JL, ri);
						ri += ns;
						if (ri <= lastbig && ns < SEG) {
#pragma unroll
							for (int q = 0; q < LQ; q++) {
								if (lane + 64 * q < MURTY_OUT) ws.logcomp[lane + 64 * q] = l[q];
							}
							lds_fence();
							replay_one(ri);
#ifdef PHD_STAMPS
							stamp_[11] += 1000;
#endif
							lds_fence();
#pragma unroll
							for (int q = 0; q < LQ; q++) l[q] = (lane + 64 * q < MURTY_OUT) ? ws.logcomp[lane + 64 * q] : 0.0;
							maxm = MURTY_OUT;
							T = 0;
#pragma unroll
							for (int q = 1; q < LQ; q++) T = fma(l[q], l[q], T);
							ri++;
						}
					}
				}
				else if (!GRAD && J >= 5) {
					// Only what the clusters of up to 5 rows LEAVE in logcomp matters to a big one (its values are known from the
					// permanent path): entry i is pairing number i of the last cluster before it that has more than i pairings
					// (a cluster of n rows writes its n! values into [0, n!), in LexicographicalPairing's order — `modelsize` = J >=
					// 5 >= n: every permutation once). So the clusters between two big ones are not replayed one after the other
					// (one lane per cluster, a hundred clusters, 1.5 M cycles for the particles that had a big cluster at all —
					// the launch waited for them): the lanes take the ENTRIES — the last cluster per row count by a wave maximum,
					// the pairing unranked, its value summed over the rows in order (AssignmentValue), as the literal walk does.
					auto small_value = [&](int ri, int idx) {
						const int root = roots[ri];
						const int cn = cnt[root];
						const int nl = cn & 0xffff, nz = cn >> 16, nrow = nl + nz;
						const unsigned long long Lp = pk_sort5(memL[root], nl), Zp = pk_sort5(memZ[root], nz);   // members, ascending
						const unsigned int perm = pk_unrank(idx, nrow);
						double v = 0;
						for (int x = 0; x < nrow; x++) {
							const int y = pk_get(perm, x);
							double e = -INFINITY;
							if (x < nl) {
								const int j = (int) ((Lp >> (12 * x)) & 4095);
								if (y < nz) {
									const int k = (int) ((Zp >> (12 * y)) & 4095);
									if ((adj[(size_t) j * MW + (k >> 6)] >> (k & 63)) & 1ull) {
										const double dist = sqrt(quad_gen(prm.Rinv, zh[j] - zs[k * 3], zh[JS + j] - zs[k * 3 + 1],
										                                  zh[2 * JS + j] - zs[k * 3 + 2]));
										e = lpd[j] + logmult - 0.5 * dist * dist;
									}
								}
								else if (y - nz == x) e = lmd[j];
							}
							else {
								if (y < nz) { if (y == x - nl) e = prm.logkappa; }
								else e = 0;
							}
							v += e;   // AssignmentValue: the rows in order
						}
						return v;
					};
					auto apply_small = [&](int lo, int hi) {   // the clusters [lo, hi), all of at most 5 rows, as logcomp sees them
						int l2 = -1, l3 = -1, l4 = -1, l5 = -1;
						for (int ri = lo + lane; ri < hi; ri += 64) {
							const int cn = cnt[roots[ri]];
							const int nrow = (cn & 0xffff) + (cn >> 16);
							l2 = (nrow == 2) ? ri : l2; l3 = (nrow == 3) ? ri : l3; l4 = (nrow == 4) ? ri : l4; l5 = (nrow == 5) ? ri : l5;
						}
#pragma unroll
						for (int o = 32; o > 0; o >>= 1) {
							l2 = max(l2, __shfl_xor(l2, o, 64)); l3 = max(l3, __shfl_xor(l3, o, 64));
							l4 = max(l4, __shfl_xor(l4, o, 64)); l5 = max(l5, __shfl_xor(l5, o, 64));
						}
						for (int idx = lane; idx < 120; idx += 64) {
							int best = l5;                        // 120 pairings
							if (idx < 24) best = max(best, l4);
							if (idx < 6) best = max(best, l3);
							if (idx < 2) best = max(best, l2);
							if (best >= 0) ws.logcomp[idx] = small_value(best, idx);
						}
						lds_fence();
					};
					int done = 0;
					for (int ri = 0; ri <= lastbig; ri++) {
						if (!isnan(res[ri])) continue;   // (uniform: res is in LDS)
						apply_small(done, ri);
						replay_one(ri);
						done = ri + 1;
					}
				}
				else {
					for (int ri = 0; ri <= lastbig; ri++) replay_one(ri);
				}
				if (GRAD && lane < 6) a.qgrad[(size_t) p * 6 + lane] = gacc;
			}
			__syncthreads();
		}
		PHD_STAMP(6);
		// total over the components: clusters holding detections, the lone landmarks (misdetection only,
		// log(1 - PD_j)) and the lone measurements (clutter, log kappa). The reference adds them in that order one
		// by one; here every thread adds its share and the shares are summed in a fixed tree.
		if (!deferred) {
			double tpart = 0;
			for (int r = tid; r < nroots; r += 256) tpart += res[r];
			for (int j = tid; j < J; j += 256) {
				bool has = false;
#pragma unroll
				for (int b = 0; b < MW; b++) has |= adj[(size_t) j * MW + b] != 0;
				if (!has) tpart += lmd[j];
			}
			for (int k = tid; k < M; k += 256) {
				if (labz[k] == J + k) tpart += prm.logkappa;
			}
			__syncthreads();
			double total = block_sum(tpart);
			if (tid == 0) s_total = total;
		}
		__syncthreads();
	}
	PHD_STAMP(7);
#ifdef PHD_STAMPS
	stamp_[11] += stamp_[0] + s_nroots;   // (diagnostic: clusters + 1000 x clusters replayed whole, in slot 11)
#endif
	PHD_STAMP_FLUSH(3, 12);
	if (tid == 0) {
		if (!deferred) a.setll[p] = s_total;
		if (!QUASI) {
			a.aJ[p]      = J;
			a.account[p] = s_ccount;
		}
	}
}

#ifndef PHD_ASSOC_WAVES
#define PHD_ASSOC_WAVES 4
#endif
template <int ZB>
__global__ __launch_bounds__(256, PHD_ASSOC_WAVES) void k_alpha_assoc(const DevParams prm, const StepBufs a, int ncap)
{
	extern __shared__ __align__(16) double smem[];
	PHD_TL_BEGIN;
	PHD_SET_PRIO(PHD_LAT_PRIO);
	alpha_assoc_body<ZB, false>(prm, a, ncap, smem);
	PHD_TL_END(3);
}

// the same with the particles that need the ordered replay (a cluster of more than 5 rows) left to k_alpha_big
template <int ZB>
__global__ __launch_bounds__(256, PHD_ASSOC_WAVES) void k_alpha_assoc_main(const DevParams prm, const StepBufs a, int ncap)
{
	extern __shared__ __align__(16) double smem[];
	alpha_assoc_body<ZB, false, false, 2, 1>(prm, a, ncap, smem);
}

// WeightAlpha's last line for every particle (PHDNavigator.cs:390-392, :335), when k_alpha_density left it open (a.defer):
// alpha = exp(set log-likelihood + density ratio), weight *= alpha
__global__ __launch_bounds__(256) void k_alpha_combine(const StepBufs a)
{
	const int p = blockIdx.x * 256 + threadIdx.x;
	if (p >= a.P) return;
	const double alpha = exp(a.setll[p] + a.ratio[p]);   // :392
	a.alpha[p] = alpha;
	bank_of(a, SEL_OUT).weights[p] = bank_of(a, SEL_IN).weights[p] * alpha;   // :335
}

// one workgroup per candidate pose (the smoother's pose x landmark x measurement batches)
#ifndef PHD_QGRAD_WAVES
#define PHD_QGRAD_WAVES 2   // waves per SIMD the register allocation aims at: the gradient replay does not fit 128 registers
#endif                      // (at four waves per SIMD it ran out of 1.3 KB of scratch per lane, 18 times the value kernel's time)
template <int ZB>
__global__ __launch_bounds__(256, PHD_QGRAD_WAVES) void k_quasi_setll_grad(const DevParams prm, const StepBufs a, int ncap)
{
	extern __shared__ __align__(16) double smem[];
	__shared__ __align__(16) double gws[QGRAD_LDS_DOUBLES];
	alpha_assoc_body<ZB, true, true>(prm, a, ncap, smem, gws);
}

template <int ZB>
__global__ __launch_bounds__(256, 4) void k_quasi_setll(const DevParams prm, const StepBufs a, int ncap)
{
	extern __shared__ __align__(16) double smem[];
	alpha_assoc_body<ZB, true>(prm, a, ncap, smem);
}

// =================================================================================================
// k_alpha_density — the two mixture-density sums of WeightAlpha (PHDNavigator.cs:381-384) and its last line:
//   alpha = exp( L(Z | map estimate, pose) + [sum_j log v_pred(m_j) - sum w_pred] - [sum_j log v_corr(m_j) - sum w_corr] )
// with v(.) the full, ungated mixture density (Map.Evaluate(point), Map.cs:192-202) and m_j the landmarks of
// the map estimate left in HBM by k_alpha_assoc. Landmark per lane, component tiles broadcast from LDS.
// =================================================================================================
#define DENS_REC 12   // gauss_record + the weight ratio of the component's surviving misdetection copy (+ 1: records stay 16-byte aligned)

#ifndef DENS_TILE
#define DENS_TILE TILE   // components staged per tile of THIS kernel (threads beyond it sit the staging out)
#endif
#ifndef PHD_DENS_WAVES
#define PHD_DENS_WAVES 4
#endif
#define DENS_LDS_DOUBLES (DENS_TILE * DENS_REC + 2 * (DENS_JL / 64) * 256 + EXPTAB_N + 2)
// WeightAlpha's last line for a particle whose density sums and set log-likelihood come from two workgroups (k_particle_chain's helper
// and main): each leaves its number (a.ratio / a.setll), waits until that store is in the L2 both sit behind and swaps the launch's
// number into the particle's ticket word; the one that finds it there already is second, reads the other's number and writes alpha
// and the weight — exp(setll + ratio), as the one-workgroup path does. One thread; returns whether it was second.
__device__ __forceinline__ bool alpha_meet(const StepBufs& a, int p, bool have_ratio, double mine)
{
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	if (atomicExch(a.dsync + 3 * (size_t) p + 2, a.dstamp) != a.dstamp) return false;
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
	const double other = __hip_atomic_load(have_ratio ? a.setll + p : a.ratio + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	const double setll = have_ratio ? other : mine, ratio = have_ratio ? mine : other;
	const double alpha = exp(setll + ratio);         // PHDNavigator.cs:392
	a.alpha[p] = alpha;
	bank_of(a, SEL_OUT).weights[p] = bank_of(a, SEL_IN).weights[p] * alpha;   // :335
	return true;
}

// What alpha_density_body can do before the step has produced anything (the helper workgroup of k_particle_chain, while it waits):
// the exponent table and the records of the PRIOR components of its first tile — they only need the prior mixture —, the
// component's weight kept in the record's spare slot. Returns the thread's share of sum w_pred so far; the body is then called
// with `pre` = true and that number and leaves those records as they are (the same arithmetic, done earlier).
__device__ __forceinline__ double alpha_density_prestage(const StepBufs& a, double* pool, int p)
{
	double* const tile = pool;
	double* const etab = tile + DENS_TILE * DENS_REC + 2 * (DENS_JL / 64) * 256;
	const int tid = threadIdx.x;
	const MixView vin = bank_view(a, SEL_IN);
	const int n = vin.count[p];
	const size_t sbi = in_base(a, p);
	exp_tab_init(etab, tid);
	double pc = 0;
	if (tid < DENS_TILE && tid < n) {
		double w, m[3], P[6], Pi[6], det;
		load_comp(vin.rec + (sbi + tid) * MIX_REC, w, m, P);
		pc = w;
		inv_sym3(P, Pi, det);
		gauss_record(w, m, Pi, PHD_INV_2PI / sqrt(fabs(det)), tile + tid * DENS_REC);
		tile[tid * DENS_REC + 11] = w;
	}
	return pc;
}

// `meet` (the helper workgroup of k_particle_chain): the body ends with alpha_meet instead of WeightAlpha's last line; returns
// whether this workgroup wrote the particle's alpha. `pre`: alpha_density_prestage has run on this pool (pre_pcount: what it returned).
__device__ __forceinline__ bool alpha_density_body(const DevParams& prm, const StepBufs& a, double* pool, int pin = -1, bool meet = false,
                                                   bool pre = false, double pre_pcount = 0.0)
{
	constexpr int JL = DENS_JL;
	double* const tile = pool;                                        // [DENS_TILE][12]
	double* const partpl = tile + DENS_TILE * DENS_REC;                    // [JB][4][64] partial densities of v_pred (HBM slab when J > JL)
	double* const partcl = partpl + (JL / 64) * 256;                  // the same for v_corr
	double* const etab = partcl + (JL / 64) * 256;
	int* const s_wc = (int*) (etab + EXPTAB_N);                       // [4]
	double* red = tile;                          // reduction scratch once the sweeps are over

	const int p = (pin >= 0) ? pin : a.p0 + blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const int cap = a.cap;
	const MixView vin = bank_view(a, SEL_IN), vout = bank_view(a, SEL_OUT);
	const Bank bin = bank_of(a, SEL_IN);
	const Bank bout = bank_of(a, SEL_OUT);
	const int n = vin.count[p], nb = a.born_count[p], no = vout.count[p];
	const int np = n + nb;
	const size_t sbi = in_base(a, p), sbo = (size_t) p * cap;
	const int J = a.aJ[p];
	const int JS = a.Jcap;
	const double* lm = a.alm + (size_t) p * 3 * JS;   // [3][Jcap]
	double* gj = a.jscratch + (size_t) p * alpha_jscratch_doubles(a.Jcap);
	double* partp = (J <= JL) ? partpl : gj + 13 * (size_t) JS;   // (the association kernel's arrays in the slab are dead by now)
	double* partc = (J <= JL) ? partcl : gj;
	const double* wcopy = a.wcopy + (size_t) p * (cap + a.Mcap);
	const int* cover = a.cover + sbo;
	if (!pre) exp_tab_init(etab, tid);
	__syncthreads();

	// ---- phase 2: sum_j log v_pred(m_j), sum_j log v_corr(m_j) with v = full ungated mixture density (Map.cs:192-202)
	// Component tiles are staged once and swept for every block of 64 landmarks (landmark per lane, the
	// component broadcast from LDS). A last block with few landmarks is packed: LJ = 2^k lanes carry the
	// landmarks and the 64 / LJ lane groups take different components, then the groups are summed.
	//
	// Most components of the corrected map are the misdetection copies of predicted components that came through
	// PruneModel alone: the same Gaussian with the weight (1 - PD) w. Their densities are not evaluated again: the sweep
	// over the predicted mixture adds ratio_c * (w_c N_c(m_j)) to the corrected sum as well, ratio_c = wcopy[c] / w_c
	// (k_prune_merge, which also checks that the copy's moments are the component's up to Merge's rounding). The second
	// sweep takes only the other corrected components (updated by a measurement, merged).
	double plog_part = 0, clog_part = 0, pcount_part = pre ? pre_pcount : 0.0;
	{
		const int JB = (J + 63) >> 6;
		// (the sum of the prior weights, sum w_pred: added up where the records are staged below — thread tid takes the
		// components tid, tid + 256, ... there, in this order)
		for (int i = tid; i < JB * 256; i += 256) { partp[i] = 0; partc[i] = 0; }
		for (int src = 0; src < 2; src++) {
			const int total = (src == 0) ? np : no;
			for (int c0 = 0; c0 < total; c0 += DENS_TILE) {
				int c = (DENS_TILE < 256 && tid >= DENS_TILE) ? total : c0 + tid;
				int cend;
				if (src == 0) {
					if (pre && c0 == 0 && c < n) {   // (staged ahead: only the copy's weight ratio is new)
						const double w = tile[tid * DENS_REC + 11];
						tile[tid * DENS_REC + 10] = (w > 0) ? wcopy[c] / w : 0.0;
					}
					else if (c < total) {
						double w, m[3], P[6], Pi[6], det;
						if (c < n) {
							load_comp(vin.rec + (sbi + c) * MIX_REC, w, m, P);
							pcount_part += w;
						}
						else {
							const double* bm = a.born_mean + ((size_t) p * a.Mcap + (c - n)) * 3;
							w = prm.birthw;
							m[0] = bm[0]; m[1] = bm[1]; m[2] = bm[2];
#pragma unroll
							for (int t = 0; t < 6; t++) P[t] = prm.birthP[t];
						}
						inv_sym3(P, Pi, det);
						gauss_record(w, m, Pi, PHD_INV_2PI / sqrt(fabs(det)), tile + tid * DENS_REC);
						tile[tid * DENS_REC + 10] = (w > 0) ? wcopy[c] / w : 0.0;
					}
					cend = min(DENS_TILE, total - c0);
				}
				else {
					// the corrected components not accounted for by the first sweep, compacted in map order
					const bool other = c < total && !cover[c];
					const unsigned long long bal = ballot64(other);
					if (lane == 0) s_wc[wv] = __popcll(bal);
					__syncthreads();
					int slot = __popcll(bal & lanemask_lt());
					for (int q = 0; q < wv; q++) slot += s_wc[q];
					cend = s_wc[0] + s_wc[1] + s_wc[2] + s_wc[3];
					if (other) {
						double w, m[3], P[6], Pi[6], det;
						load_comp(vout.rec + (sbo + c) * MIX_REC, w, m, P);
						inv_sym3(P, Pi, det);
						gauss_record(w, m, Pi, PHD_INV_2PI / sqrt(fabs(det)), tile + slot * DENS_REC);
					}
				}
				__syncthreads();
				for (int jb = 0; jb < JB && cend > 0; jb++) {
					const int rem = min(64, J - jb * 64);
					const int LJ  = (rem > 32) ? 64 : ((rem <= 1) ? 1 : (1 << (32 - __clz(rem - 1))));
					const int G   = 64 / LJ, g = lane / LJ, jl = lane & (LJ - 1);
					const bool jv = jl < rem;
					const int  j  = jb * 64 + jl;
					const double x0 = jv ? lm[j] : 0, x1 = jv ? lm[JS + j] : 0, x2 = jv ? lm[2 * JS + j] : 0;
					// w * (mult * exp(-d^T Pinv d / 2)) of component cc at this lane's landmark (Map.cs:198)
					auto dens = [&](int cc) {
						const double* tt = tile + cc * DENS_REC;
						return exp_pair(gauss_logw(tt, x0 - tt[0], x1 - tt[1], x2 - tt[2]), etab);
					};
					double acc = 0, acc2 = 0, cacc = 0, cacc2 = 0;
					int cc = wv * G + g;
					const int step = 4 * G;
					if (src == 0) {
						for (; cc + step < cend; cc += 2 * step) {   // two independent components per trip
							const double e1 = dens(cc), e2 = dens(cc + step);
							acc  += e1;
							acc2 += e2;
							cacc  = fma(tile[cc * DENS_REC + 10], e1, cacc);
							cacc2 = fma(tile[(cc + step) * DENS_REC + 10], e2, cacc2);
						}
						if (cc < cend) {
							const double e1 = dens(cc);
							acc += e1;
							cacc = fma(tile[cc * DENS_REC + 10], e1, cacc);
						}
						partp[(jb * 4 + wv) * 64 + lane] += acc + acc2;   // own slot
						partc[(jb * 4 + wv) * 64 + lane] += cacc + cacc2;
					}
					else {
						for (; cc + step < cend; cc += 2 * step) {
							acc  += dens(cc);
							acc2 += dens(cc + step);
						}
						if (cc < cend) acc += dens(cc);
						partc[(jb * 4 + wv) * 64 + lane] += acc + acc2;
					}
				}
				__syncthreads();
			}
			__threadfence_block();
			__syncthreads();
			const double* part = (src == 0) ? partp : partc;
			for (int jb = wv; jb < JB; jb += 4) {
				const int rem = min(64, J - jb * 64);
				const int LJ  = (rem > 32) ? 64 : ((rem <= 1) ? 1 : (1 << (32 - __clz(rem - 1))));
				const int jl = lane & (LJ - 1);
				double v = part[(jb * 4 + 0) * 64 + lane] + part[(jb * 4 + 1) * 64 + lane] + part[(jb * 4 + 2) * 64 + lane] +
				           part[(jb * 4 + 3) * 64 + lane];
				for (int o = LJ; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
				if (lane < LJ && jl < rem) {
					if (src == 0) plog_part += log(v);
					else          clog_part += log(v);
				}
			}
			__syncthreads();
		}
	}
	// block reductions (fixed order)
	auto block_sum = [&](double v) {
		red[tid] = v;
		__syncthreads();
		for (int s = 128; s > 0; s >>= 1) {
			if (tid < s) red[tid] += red[tid + s];
			__syncthreads();
		}
		double r = red[0];
		__syncthreads();
		return r;
	};
	const double plog = block_sum(plog_part);
	const double clog = block_sum(clog_part);
	const double pcount = block_sum(pcount_part) + nb * prm.birthw;
	if (tid == 0) {
		const double ccount = a.account[p];
		double ratio = (plog - pcount) - (clog - ccount);   // :390
		if (meet) {
			__hip_atomic_store(a.ratio + p, ratio, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			s_wc[0] = alpha_meet(a, p, true, ratio) ? 1 : 0;
		}
		else if (a.defer) a.ratio[p] = ratio;               // (k_alpha_combine: the set log-likelihood may still be in the making)
		else {
			double alpha = exp(a.setll[p] + ratio);         // :392
			a.alpha[p] = alpha;
			const double wnew = bin.weights[p] * alpha;     // :335
			// (tickets: written through to memory — k_normalise_resample on the other stream reads it without a launch boundary between)
			if (a.tickets) __hip_atomic_store(&bout.weights[p], wnew, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			else bout.weights[p] = wnew;
		}
	}
	if (!meet) return true;
	__syncthreads();
	return s_wc[0] != 0;
}

__global__ __launch_bounds__(256, PHD_DENS_WAVES) void k_alpha_density(const DevParams prm, const StepBufs a)
{
	__shared__ __align__(16) double pool[DENS_LDS_DOUBLES];
	PHD_TL_BEGIN;
	PHD_SET_PRIO(PHD_DENSE_PRIO);
	alpha_density_body(prm, a, pool);
	// (device-side ordering of the sub-range streams, phd_step_async: thread 0 wrote the particle's weight through to memory; once
	// that store is acknowledged it takes the ticket k_normalise_resample counts. No fence: a device-scope release would write
	// back the XCD's whole L2 — the other stream's kernels' lines with it — 2048 times a step)
	if (a.tickets && threadIdx.x == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); atomicAdd(a.ticket, 1u); }
	PHD_TL_END(4);
}

// ... and those particles, INSIDE the launch of the densities: the first `nbig` workgroups of k_alpha_density_big stride over the
// list (how many there are is known on the device only) and run the association body again with the replay; the others are
// k_alpha_density's. A replay is one wave deep in the solver for hundreds of microseconds: started first, beside a launch
// that fills the machine for as long, it costs the step nothing. (On a stream of its own it did not overlap: HIP's streams
// share four hardware queues, and with more of those the whole step ran a fifth slower.)
template <int ZB>
__global__ __launch_bounds__(256, 4) void k_alpha_density_big(const DevParams prm, const StepBufs a, int ncap, int nbig)
{
	extern __shared__ __align__(16) double smem[];
	if ((int) blockIdx.x < nbig) {
		const int n = a.biglist[0];
		for (int w = blockIdx.x; w < n; w += nbig) {
			alpha_assoc_body<ZB, false, false, 0, 2>(prm, a, ncap, smem, nullptr, a.biglist[1 + w]);
			__syncthreads();   // (the next particle reuses the LDS arrays)
		}
		return;
	}
	alpha_density_body(prm, a, smem, a.p0 + (int) blockIdx.x - nbig);
}
