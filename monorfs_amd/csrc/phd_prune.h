// phd_prune.h — k_prune_merge: PruneModel (PHDNavigator.cs:913-948) for one particle per workgroup.
//
// The reference sorts the corrected mixture by weight, keeps the first min(MaxQuantity, #w >= MinWeight)
// entries and then, in weight order, lets every still-present entry i absorb every later entry k with
// (m_i - m_k)^T P_i^-1 (m_i - m_k) < MergeThreshold^2 (Gaussian.AreClose, Gaussian.cs:243-246), replacing
// the set by its moment match (Gaussian.Merge, Gaussian.cs:297-347).
//
// Only the question "is i still present" is sequential. The kernel therefore splits the work:
//   A. sort     : bitonic sort on the weight's bit pattern (one 64-bit word per entry: key + emit slot), held in
//                 registers (thread exchanges, lane shuffles, LDS only for the longest distances), the best
//                 half kept while further chunks stream in; runs of equal weights are then put in
//                 canonical-index order, which makes the order the reference's sort made stable
//   S. stage    : the kept records are gathered once, in sorted order, into a slab of 96-byte rows in HBM (the
//                 component record, as the banks and the emitted list hold it, + its canonical index: a gathered row
//                 is two 64-byte sectors whatever its origin, a row read back is one or two lines); the Euclidean
//                 bounds go to LDS, the means of a thread's rows stay in its registers for the grid
//   B. pairs    : every closeness test close_i(k), k > i, in parallel, one row per thread. A Euclidean bound
//                 |d|^2 > T^2 trace(P_i)  =>  d^T P_i^-1 d >= |d|^2 / lambda_max(P_i) > T^2
//                 limits the candidates of a row to a ball; the rows are binned in a uniform grid (hashed
//                 cells) and the <= 8 buckets the ball meets are walked as one list of float32 records, the
//                 survivors of the float32 distance test take the exact FP64 test. Each row keeps its first 7
//                 close rows. The rows go to the lanes in the order of their candidate counts (a second counting
//                 sort), so that a wave is not held up by the one row of a crowded cell in it.
//   C. resolve  : one wave walks the rows in weight order, 64 row records at a time held in its lanes and
//                 the "absorbed" bits spread over its lanes (integer work only); rows with more than 7
//                 close rows are re-tested by the 64 lanes.
//   D. merge    : one thread per surviving row accumulates the raw moments of its set in list order
//                 (leader first, members by rank — the reference's summation order) and writes the
//                 result at its position among the survivors.
#pragma once
#include "phd_device.h"

#define PRUNE_NBR 7
#ifndef PRUNE_NB
#define PRUNE_NB 2048   // buckets of the spatial hash (16-bit counters, two per LDS word; 4096 until round 4: half of them, and the bounds below as
#endif                  // float32, take MaxQuantity 1024's working set from 58.7 to 50.5 KB — three workgroups per CU instead of two on config S)
#ifndef PRUNE_HEAVY
#define PRUNE_HEAVY 32  // a row with more candidates than this is searched by a whole wave
#endif
#define PRUNE_ROW 12    // doubles per row of the sorted slab: w, m[3], P[6] (a component record), canonical index, spare

struct PruneLds {
	int rad2, x, scan;       // offsets in doubles
	int NS;                  // sort width (power of two >= 2 * cutcap)
	int bytes;
};

__host__ __device__ inline PruneLds prune_lds(int cutcap)
{
	PruneLds l;
	int cc = (cutcap + 1) & ~1;
	int NS = 512;
	while (NS < 2 * cutcap) NS <<= 1;
	l.NS    = NS;
	l.rad2  = 0;
	l.x     = l.rad2 + ((cc / 2 + 1) & ~1);   // (rad2: cc float32) sort words u64[NS]  |  the lists of the pair search (see `rest`)
	// nbr u64[2*cc], cand float4[cc], owner int[cc], cstart u16[NB+2], absb int[64], ord u16[cc], rowpos u16[cc]
	int rest  = 2 * cc + 2 * cc + cc / 2 + (PRUNE_NB + 2) / 4 + 1 + 32 + 2 * (cc / 4 + 1) + 4;
	int need  = NS > rest ? NS : rest;
	const int ranked = NS + 516 + (NS * 2) / 3 + 2;   // what the ranked ordering takes with its canonical indices in LDS (prune_merge_body, haveki)
	if (ranked > need) need = ranked;
	l.scan  = l.x + need;
	l.bytes = (l.scan + 136) * 8;            // int[264] | double[28], + spare
	return l;
}

// total order on doubles as unsigned integers (larger double <-> larger key); 0 is below every number
__device__ __forceinline__ unsigned long long prune_key(double w)
{
	unsigned long long b = (unsigned long long) __double_as_longlong(w);
	return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

// Sort entries are one 64-bit word: the top 44 bits of the weight's key (sign, exponent, 32 mantissa bits),
// then a 20-bit payload (slot). Entries whose weights agree in those 44 bits (relative difference below
// 2.4e-10) come out adjacent in arbitrary order; the caller then orders each such run by (full weight desc,
// canonical index asc) — the reference's sort made stable. 0 sorts last (padding).
#define PRUNE_SLOT_BITS 20
#define PRUNE_SLOT_MASK ((1u << PRUNE_SLOT_BITS) - 1u)
__device__ __forceinline__ unsigned long long prune_pack(double w, int slot)
{
	return (prune_key(w) & ~(unsigned long long) PRUNE_SLOT_MASK) | (unsigned int) slot;
}
__device__ __forceinline__ int prune_slot(unsigned long long v) { return (int) ((unsigned int) v & PRUNE_SLOT_MASK); }
__device__ __forceinline__ unsigned long long prune_kbits(unsigned long long v) { return v >> PRUNE_SLOT_BITS; }

template <class W>
__device__ __forceinline__ void prune_ce(W* v, int t, int j, int k)
{
	int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
	int q = i | j;
	W a = v[i], b = v[q];
	bool desc = (i & k) == 0;
	if (desc ? (a < b) : (a > b)) { v[i] = b; v[q] = a; }
}

// bitonic sort of n (power of two) words, largest first, by 256 threads. Compare-exchange t belongs to the
// 128-element block t / 64 for every distance j <= 64, and a wave keeps the same blocks from one distance to
// the next, so those sub-steps need no workgroup barrier — only the distances >= 128 and the stage ends do.
template <class W>
__device__ __forceinline__ void prune_bitonic(W* v, int n, int tid)
{
	for (int k = 2; k <= n; k <<= 1) {
		int j = k >> 1;
		for (; j >= 128; j >>= 1) {
			for (int t = tid; t < (n >> 1); t += 256) prune_ce(v, t, j, k);
			__syncthreads();
		}
		for (; j > 0; j >>= 1) {
			for (int t = tid; t < (n >> 1); t += 256) prune_ce(v, t, j, k);
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
			__builtin_amdgcn_wave_barrier();
		}
		__syncthreads();
	}
}

// In-register bitonic sort of n = 256 * EPT words, largest first: thread t holds the EPT consecutive words
// sv[t * EPT ..). Distances below EPT are exchanges inside a thread, distances below 64 * EPT are lane
// shuffles inside a wave; only the few larger ones go through LDS.
template <int EPT, class W>
__device__ __forceinline__ void prune_sort_regs(W* sv, int tid)
{
	constexpr int n = 256 * EPT;
	W v[EPT];
#pragma unroll
	for (int i = 0; i < EPT; i++) v[i] = sv[tid * EPT + i];
	for (int k = 2; k <= n; k <<= 1) {
		int j = k >> 1;
		if (j >= 64 * EPT) {
#pragma unroll
			for (int i = 0; i < EPT; i++) sv[tid * EPT + i] = v[i];
			__syncthreads();
			for (; j >= 64 * EPT; j >>= 1) {
				for (int t = tid; t < (n >> 1); t += 256) prune_ce(sv, t, j, k);
				__syncthreads();
			}
#pragma unroll
			for (int i = 0; i < EPT; i++) v[i] = sv[tid * EPT + i];
		}
		for (; j >= EPT; j >>= 1) {
			const int lx = j / EPT;
			const bool lower = (tid & lx) == 0;
#pragma unroll
			for (int i = 0; i < EPT; i++) {
				const bool desc = ((tid * EPT + i) & k) == 0;
				const W o = __shfl_xor(v[i], lx, 64);
				const W hi = v[i] > o ? v[i] : o, lo = v[i] > o ? o : v[i];
				v[i] = (lower == desc) ? hi : lo;
			}
		}
#pragma unroll
		for (int jj = EPT / 2; jj > 0; jj >>= 1) {
			if (jj <= (k >> 1)) {
#pragma unroll
				for (int i = 0; i < EPT; i++) {
					if ((i & jj) == 0) {
						const bool desc = ((tid * EPT + i) & k) == 0;
						const W x = v[i], y = v[i | jj];
						const W hi = x > y ? x : y, lo = x > y ? y : x;
						v[i]      = desc ? hi : lo;
						v[i | jj] = desc ? lo : hi;
					}
				}
			}
		}
	}
#pragma unroll
	for (int i = 0; i < EPT; i++) sv[tid * EPT + i] = v[i];
	__syncthreads();
}

template <class W>
__device__ __forceinline__ void prune_sort(W* sv, int n, int tid)
{
	switch (n) {
	case 256:  prune_sort_regs<1>(sv, tid); break;
	case 512:  prune_sort_regs<2>(sv, tid); break;
	case 1024: prune_sort_regs<4>(sv, tid); break;
	case 2048: prune_sort_regs<8>(sv, tid); break;
	case 4096: prune_sort_regs<16>(sv, tid); break;
	default:   prune_bitonic(sv, n, tid);
	}
}

// The sort words of k_prune_merge are 32 bits wide: the top bits of a (positive) weight — exponent and leading mantissa
// bits, as many as the slot leaves: 22 for up to 1024 emitted components, relative difference below 1e-3 inside a key —
// then the slot. Half the shuffles and compares of the 64-bit words above; the runs of equal keys are a little longer
// (they are ordered by the full weights afterwards, as there). 0 sorts last (padding); no real word is 0.
__device__ __forceinline__ unsigned int prune_pack32(double w, int slot, int sbits)
{
	const int kbn = 32 - sbits;
	unsigned int kb = (unsigned int) (((unsigned long long) __double_as_longlong(w) << 1) >> (64 - kbn));
	const unsigned int top = (1u << kbn) - 2u;
	kb = (kb > top ? top : kb) + 1u;
	return (kb << sbits) | (unsigned int) slot;
}
// the 32 bits of the weight behind those of the key
__device__ __forceinline__ unsigned int prune_next32(double w, int sbits)
{
	return (unsigned int) ((((unsigned long long) __double_as_longlong(w) << 1) << (32 - sbits)) >> 32);
}

__device__ __forceinline__ void prune_merge_body(const DevParams& prm, const StepBufs& a, int cutcap, double* smem, long long tk0 = 0)
{
	const PruneLds lay = prune_lds(cutcap);
	const int cc = (cutcap + 1) & ~1, NS = lay.NS;
	float*  rad2  = (float*) (smem + lay.rad2);            // [cut] squared Euclidean bound of row i (inf: none), float32 rounded UP: a sound bound stays one
	unsigned int* sw = (unsigned int*) (smem + lay.x);                 // [NS] sort words (prune_pack32)
	unsigned int* w2 = sw + NS;                                        // [NS] the 32 weight bits behind the key of slot e < NS
	unsigned long long* nbr = (unsigned long long*) (smem + lay.x);    // [cut][2]: count + up to 7 close later rows
	float4* cand  = (float4*) (nbr + 2 * cc);              // [cut] rows grouped by bucket: mean relative to the box (float32), row
	int*    owner = (int*) (cand + cc);                    // [cut] row that absorbed k (-1: none); the row's bucket while the grid is built
	unsigned int* cw = (unsigned int*) (owner + cc);       // [(NB + 2) / 2] two 16-bit bucket boundaries per word
	const unsigned short* cstart = (const unsigned short*) cw;   // [NB + 2] bucket b is [cstart[b], cstart[b + 1])
	int*    absb  = (int*) (cw + (PRUNE_NB + 2) / 2 + 1);  // [64] absorbed bits as left by the resolving wave
	unsigned short* ord = (unsigned short*) (absb + 64);   // [cut] rows in the order the pair search takes them
	unsigned short* rowpos = ord + cc;                      // [cut] position of row r in `cand`
	int*    scan  = (int*) (smem + lay.scan);              // [264]
	double* bred  = smem + lay.scan;                       // [28] block reduction scratch (before `scan` is used)

	// (the particle's number through an empty asm: in the fused launch the compiler otherwise computes this body's addresses at the
	// top of the kernel and carries them — spilled — across the emit body)
	int p_ = a.p0 + blockIdx.x;
	asm volatile("" : "+s"(p_));
	const int p = p_, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	PHD_STAMP_DECL;
	const MixView vout = bank_view(a, SEL_OUT);
	const int ne = a.emit_count[p];
	const size_t eb = (size_t) p * a.ecap;
	const double merge_thr2 = prm.merge_thr2;   // a local copy: a lambda that captured `prm` by reference would pin the argument block to memory
	const int cut = min(min(prm.maxq, ne), cutcap);   // weightcut: every emitted weight is already >= MinWeight
	double* srec = a.srec + (size_t) p * PRUNE_ROW * cutcap;  // [cutcap][12] the kept records in sorted order: component record (w, mean, covariance), canonical index, spare

	PHD_STAMP(0);
	// ---- A. order by (weight desc, canonical index asc)
	int sbits = 1;
	while ((1 << sbits) < ne) sbits++;                    // bits of a slot number
	const unsigned int smask = (1u << sbits) - 1u;
	const unsigned int* order = sw;                       // the kept entries' slots in sorted order (low `sbits` bits)
	// ranked path: order[r] is the entry's place in the scatter arrays instead, where its key (= its weight), its slot and
	// (rk_ki != NULL) its canonical index are — the gather then needs no trip to memory before the record's own
	const unsigned long long* rk_kw = nullptr;
	const unsigned int* rk_ks = nullptr;
	const unsigned int* rk_ki = nullptr;
	{
		int n = 256;                                      // (at least the width of the register sort: small maps, too, skip the barrier-per-stage version)
		while (n < ne && n < NS) n <<= 1;                 // width of the first pass
		int taken = 0;                                    // emitted entries consumed so far
		bool first = true, prefilled = false;
		// Only the `cut` heaviest entries are kept. A histogram over the weights' leading bits (exponent and five mantissa
		// bits, counted from MinWeight up: every emitted weight is at least that) finds the bin the cut-th heaviest entry lies
		// in; whole bins are taken, so nothing changes in the order of what is kept.
		//   * ranked: the entries from that bin up are scattered to their bins' places (a counting sort: the bins' starts are
		//     the suffix sums of the histogram) and every entry counts, among the few entries of its own bin, those that
		//     go before it — by the full weight, then the canonical index: the reference's sort made stable, with no sorting
		//     network, no runs of equal keys to order afterwards;
		//   * when that does not fit (more entries than the LDS arrays hold, or a bin crowded with hundreds of equal weights):
		//     the entries from the threshold bin up are sorted in one pass of the smallest width (more than 1024 emitted), or
		//     everything goes through the passes below.
		const int xsize = lay.scan - lay.x;
		const int capN = (NS * 2) / 3;                                 // entries the ranked path holds: 12 bytes each in the sort arrays' place
		bool ranked = false;
		if (ne > 256 && cut > 0 && xsize >= NS + 516 + capN / 2 + 2) {
			int* const hist = (int*) (smem + lay.x + NS);   // [1024] behind the sort arrays: counts, then starts, then ends of the bins
			int& s_T = hist[1024];
			int& s_cnt = hist[1025];
			int& s_fill = hist[1026];
			int& s_maxb = hist[1027];
			unsigned int* const out = (unsigned int*) (hist + 1032);   // [capN]
			const unsigned int base = (unsigned int) (((unsigned long long) __double_as_longlong(prm.minw) << 1) >> 48);
			auto bin_of_bits = [&](unsigned long long bits) {          // bits of a positive double, or its prune_key (the top bit falls out)
				const unsigned int k = (unsigned int) ((bits << 1) >> 48);
				return (int) min(max((int) k - (int) base, 0), 1023);
			};
			auto bin_of = [&](double w) { return bin_of_bits((unsigned long long) __double_as_longlong(w)); };
			for (int t = tid; t < 1028; t += 256) hist[t] = 0;
			__syncthreads();
			// (up to 2048 emitted entries a thread keeps its weights for the scatter below: one trip to memory instead of two)
			constexpr int WK = 8;
			const bool keepw = ne <= 256 * WK;
			// (their canonical indices — which decide between equal weights: components never detected since birth, dozens of
			// them — are kept beside the slots when the pool has the room, and fetched HERE, with the weights, in one trip: read
			// one by one inside the scatter below they were a dependent trip to memory per entry, five in a row per thread)
			const bool haveki = xsize >= NS + 516 + capN + 2;
			double wk[WK];
			int    ik[WK];
#pragma unroll
			for (int q = 0; q < WK; q++) {
				const int e = tid + 256 * q;
				wk[q] = (keepw && e < ne) ? a.emit_w[eb + e] : 0.0;
				ik[q] = (keepw && haveki && e < ne) ? a.emit_idx[eb + e] : 0;
			}
			if (keepw) {
#pragma unroll
				for (int q = 0; q < WK; q++) {
					if (tid + 256 * q < ne) atomicAdd(&hist[bin_of(wk[q])], 1);
				}
			}
			else {
				for (int e0 = tid; e0 < ne; e0 += 256 * WK) {
					double wr[WK];
#pragma unroll
					for (int q = 0; q < WK; q++) wr[q] = (e0 + 256 * q < ne) ? a.emit_w[eb + e0 + 256 * q] : 0.0;
#pragma unroll
					for (int q = 0; q < WK; q++) {
						if (e0 + 256 * q < ne) atomicAdd(&hist[bin_of(wr[q])], 1);
					}
				}
			}
			__syncthreads();
			if (wv == 0) {
				// from the top bin down (lane l: bins 1023 - 16 l .. 1008 - 16 l): every bin's count gives way to its start, the
				// number of entries in the bins above it; the first bin at which `cut` entries are reached is the threshold
				int cq[16], mine = 0;
#pragma unroll
				for (int q = 0; q < 16; q++) { cq[q] = hist[1023 - (16 * lane + q)]; mine += cq[q]; }
				int incl = mine;
#pragma unroll
				for (int o = 1; o < 64; o <<= 1) {
					const int y = __shfl_up(incl, o, 64);
					if (lane >= o) incl += y;
				}
				int run = incl - mine, mxb = 0;
#pragma unroll
				for (int q = 0; q < 16; q++) {
					const int bq = 1023 - (16 * lane + q);
					hist[bq] = run;
					if (run < cut) {
						mxb = max(mxb, cq[q]);
						if (cut <= run + cq[q]) { s_T = bq; s_cnt = run + cq[q]; }
					}
					run += cq[q];
				}
#pragma unroll
				for (int o = 32; o > 0; o >>= 1) mxb = max(mxb, __shfl_xor(mxb, o, 64));
				if (lane == 0) s_maxb = mxb;
			}
			__syncthreads();
			const int T = s_T, cnt = s_cnt;
			if (cnt <= capN && s_maxb <= 512) {
				unsigned long long* const kw = (unsigned long long*) (smem + lay.x);   // [capN] full keys, bin after bin
				unsigned int* const ks = (unsigned int*) (kw + capN);                  // [capN] their slots
				unsigned int* const ki = out + capN;                                   // [capN] their canonical indices (haveki; read from memory pair by pair otherwise)
				auto place = [&](double w, int e, int idx) {
					const int bq = bin_of(w);
					if (bq >= T) {
						const int pos = atomicAdd(&hist[bq], 1);   // (the start becomes the end)
						kw[pos] = prune_key(w);
						ks[pos] = (unsigned int) e;
						if (haveki) ki[pos] = (unsigned int) idx;
					}
				};
				if (keepw) {
#pragma unroll
					for (int q = 0; q < WK; q++) {
						if (tid + 256 * q < ne) place(wk[q], tid + 256 * q, ik[q]);
					}
				}
				else {
					// more than 2048 emitted entries: the same in rounds of WK entries per thread, a round's loads issued together
					for (int e0 = tid; e0 < ne; e0 += 256 * WK) {
#pragma unroll
						for (int q = 0; q < WK; q++) {
							const int e = e0 + 256 * q;
							wk[q] = (e < ne) ? a.emit_w[eb + e] : 0.0;
							ik[q] = (haveki && e < ne) ? a.emit_idx[eb + e] : 0;
						}
#pragma unroll
						for (int q = 0; q < WK; q++) {
							if (e0 + 256 * q < ne) place(wk[q], e0 + 256 * q, ik[q]);
						}
					}
				}
				__syncthreads();
				PHD_STAMP(6);
				for (int pos = tid; pos < cnt; pos += 256) {
					const unsigned long long key = kw[pos];
					const int bq = bin_of_bits(key);
					const int f0 = (bq == 1023) ? 0 : hist[bq + 1], f1 = hist[bq];
					const unsigned int myslot = ks[pos];
					int rank = 0, ties = 0;
					for (int f = f0; f < f1; f += 4) {   // four bin-mates per trip, their keys fetched together
						unsigned long long kf[4];
#pragma unroll
						for (int q = 0; q < 4; q++) kf[q] = kw[min(f + q, f1 - 1)];
#pragma unroll
						for (int q = 0; q < 4; q++) {
							const bool in = f + q < f1;
							rank += (in && kf[q] > key) ? 1 : 0;
							ties += (in && kf[q] == key) ? 1 : 0;
						}
					}
					if (ties > 1) {   // others with this very weight: the canonical index decides
						const int myidx = haveki ? (int) ki[pos] : a.emit_idx[eb + myslot];
						for (int f = f0; f < f1; f++) {
							if (f != pos && kw[f] == key) {
								const int fidx = haveki ? (int) ki[f] : a.emit_idx[eb + ks[f]];
								if (fidx < myidx) rank++;
							}
						}
					}
					out[f0 + rank] = (unsigned int) pos;
				}
				__syncthreads();
				PHD_STAMP(11);
				order = out;
				rk_kw = kw; rk_ks = ks; rk_ki = haveki ? ki : nullptr;
				ranked = true;
				taken = ne;
			}
			else if (ne > 1024 && cnt <= NS) {   // (else: a crowd of equal weights at the cut — the passes below take everything)
				n = 256;
				while (n < cnt) n <<= 1;
				for (int e = tid; e < ne; e += 256) {
					const double w = a.emit_w[eb + e];
					if (e < NS) w2[e] = prune_next32(w, sbits);
					if (bin_of(w) >= T) sw[atomicAdd(&s_fill, 1)] = prune_pack32(w, e, sbits);
				}
				for (int t = cnt + tid; t < n; t += 256) sw[t] = 0u;
				taken = ne;
				prefilled = true;
			}
		}
		while (!ranked && (first || taken < ne)) {
			// first pass: fill [0, n); later passes (n == NS): refill the worse half [n/2, n)
			const int from = first ? 0 : (n >> 1);
			for (int t = from + tid; t < n && !prefilled; t += 256) {
				int e = taken + (t - from);
				unsigned int word = 0u;
				if (e < ne) {
					const double w = a.emit_w[eb + e];
					word = prune_pack32(w, e, sbits);
					if (e < NS) w2[e] = prune_next32(w, sbits);
				}
				sw[t] = word;
			}
			if (!prefilled) taken += n - from;
			first = false;
			__syncthreads();
			PHD_STAMP(6);    // (the last pass: fill | sort | runs)
			prune_sort(sw, n, tid);
			PHD_STAMP(11);
			// runs that agree in the key bits: order by (weight desc, canonical index asc); the thread at the
			// head of a run sorts it (runs are disjoint)
			for (int r = tid; r + 1 < n; r += 256) {
				const unsigned int kb = sw[r] >> sbits;
				if (sw[r + 1] == 0u || (sw[r + 1] >> sbits) != kb || (r > 0 && (sw[r - 1] >> sbits) == kb)) continue;
				int e = r + 2;
				while (e < n && sw[e] != 0u && (sw[e] >> sbits) == kb) e++;
				// (the 32 weight bits behind the key decide nearly every pair: they are in LDS; the full weights and the
				// canonical indices, a dependent HBM access each, only behind a tie in all 54 bits)
				auto next32 = [&](int sl) { return sl < NS ? w2[sl] : prune_next32(a.emit_w[eb + sl], sbits); };
				for (int x = r + 1; x < e; x++) {   // insertion sort of the run
					const unsigned int vx = sw[x];
					const int sx = (int) (vx & smask);
					const unsigned int nx = next32(sx);
					int y = x - 1;
					while (y >= r) {
						const int sy = (int) (sw[y] & smask);
						const unsigned int ny = next32(sy);
						bool stays = ny > nx;   // sy stays before sx
						if (ny == nx) {
							const double wy = a.emit_w[eb + sy], wx = a.emit_w[eb + sx];
							stays = wy > wx || (wy == wx && a.emit_idx[eb + sy] < a.emit_idx[eb + sx]);
						}
						if (stays) break;
						sw[y + 1] = sw[y];
						y--;
					}
					sw[y + 1] = vx;
				}
			}
			__syncthreads();
		}
	}
	PHD_STAMP(1);
	// ---- the kept records, gathered once into sorted order (HBM, plane per field; every later read of a row's own
	// record is coalesced); the Euclidean bounds go to LDS.
	// A pair can only be close when |m_i - m_k|^2 <= T^2 trace(P_i) (d^T P_i^-1 d >= |d|^2 / lambda_max(P_i) >=
	// |d|^2 / trace(P_i) for a positive definite P_i).
	double lo0 = INFINITY, lo1 = INFINITY, lo2 = INFINITY, hi0 = -INFINITY, hi1 = -INFINITY, hi2 = -INFINITY, rmx = 0;
	const MixView vpre = bank_view(a, SEL_IN);
	const int nprior = vpre.count[p], npredicted = nprior + a.born_count[p];
	const double* prior = vpre.rec + in_base(a, p) * MIX_REC;
	// Row r of the sorted order: where it comes from. A misdetection copy (canonical index below the size of the predicted
	// mixture, PHDNavigator.cs:837-840) IS the predicted component with another weight: its record is read where it is, in the
	// prior bank (a birth's is made up); a detection update's record is the one k_emit_finish wrote. Either is one 80-byte
	// record. First the rows' descriptions (LDS on the ranked path, the emitted list otherwise), then their records: the
	// two trips of a thread's rows overlap.
	double birthP[6];
#pragma unroll
	for (int t = 0; t < 6; t++) birthP[t] = prm.birthP[t];   // (a local copy, as merge_thr2: the lambdas below must not capture `prm`)
	// The gather is written WITHOUT control flow between a thread's loads: behind a divergent branch the compiler waits for
	// every outstanding memory operation (s_waitcnt vmcnt(0) at the join — it cannot count them across the branch), which made
	// a thread's rows one full trip to memory each, one after the other. Rows beyond the cut are clamped to the last kept row
	// (their loads hit lines the wave reads anyway, their stores are predicated off); a birth's copy reads its (unwritten)
	// emit record like an update and is patched afterwards (rare: the one branch, behind all loads).
	struct RowSrc { const double* rec; double w; int cidx; };
	auto resolve_ranked = [&](int r) {
		RowSrc o;
		const unsigned int pos = order[r];
		const int slt = (int) rk_ks[pos];
		o.w = __longlong_as_double((long long) (rk_kw[pos] & 0x7fffffffffffffffull));   // prune_key of a positive weight, undone
		o.cidx = (int) rk_ki[pos];
		o.rec = (o.cidx < nprior) ? prior + (size_t) o.cidx * MIX_REC : a.emit_rec + (eb + slt) * MIX_REC;
		return o;
	};
	auto resolve_listed = [&](int slt, double w, int cidx) {
		RowSrc o;
		o.w = w; o.cidx = cidx;
		o.rec = (cidx < nprior) ? prior + (size_t) cidx * MIX_REC : a.emit_rec + (eb + slt) * MIX_REC;
		return o;
	};
	struct RowRec { double m0, m1, m2, P0, P1, P2, P3, P4, P5; };
	auto fetch = [&](const RowSrc& o) {   // (five 16-byte loads; the weight in front of the mean is the entry's own, o.w)
		const double2* q = (const double2*) o.rec;
		const double2 a0 = q[0], a1 = q[1], a2 = q[2], a3 = q[3], a4 = q[4];
		return RowRec{a0.y, a1.x, a1.y, a2.x, a2.y, a3.x, a3.y, a4.x, a4.y};
	};
	// the copy of a birth (canonical index in [nprior, npredicted)): mean from the sweep, BirthCovariance — selects, no branch
	auto patch_birth = [&](const RowSrc& o, RowRec v) {
		const bool birth = o.cidx >= nprior && o.cidx < npredicted;
		const double* bm = a.born_mean + ((size_t) p * a.Mcap + (birth ? o.cidx - nprior : 0)) * 3;
		const double b0 = bm[0], b1 = bm[1], b2 = bm[2];
		v.m0 = birth ? b0 : v.m0; v.m1 = birth ? b1 : v.m1; v.m2 = birth ? b2 : v.m2;
		v.P0 = birth ? birthP[0] : v.P0; v.P1 = birth ? birthP[1] : v.P1; v.P2 = birth ? birthP[2] : v.P2;
		v.P3 = birth ? birthP[3] : v.P3; v.P4 = birth ? birthP[4] : v.P4; v.P5 = birth ? birthP[5] : v.P5;
		return v;
	};
	// ... its record into the slab; its bound into LDS; the box and the largest radius
	auto stage = [&](int r, const RowSrc& o, const RowRec& v) {
		double2* row = (double2*) (srec + (size_t) r * PRUNE_ROW);
		row[0] = make_double2(o.w, v.m0); row[1] = make_double2(v.m1, v.m2);
		row[2] = make_double2(v.P0, v.P1); row[3] = make_double2(v.P2, v.P3); row[4] = make_double2(v.P4, v.P5);
		row[5] = make_double2((double) o.cidx, 0.0);
		const double det = v.P0 * (v.P3 * v.P5 - v.P4 * v.P4) - v.P1 * (v.P1 * v.P5 - v.P4 * v.P2) + v.P2 * (v.P1 * v.P4 - v.P3 * v.P2);
		const bool pd = v.P0 > 0 && (v.P0 * v.P3 - v.P1 * v.P1) > 0 && det > 0;   // Sylvester
		const double rad = pd ? sqrt(merge_thr2 * (v.P0 + v.P3 + v.P5)) : INFINITY;
		{
			const double b2 = rad * rad * (1.0 + 1e-6);
			float bf = (float) b2;
			if ((double) bf < b2) bf = __int_as_float(__float_as_int(bf) + 1);   // (positive, finite: the next float32 up; +inf converts to +inf)
			rad2[r] = bf;
		}
		rmx = pd ? fmax(rmx, rad) : rmx;
		lo0 = fmin(lo0, v.m0); lo1 = fmin(lo1, v.m1); lo2 = fmin(lo2, v.m2);
		hi0 = fmax(hi0, v.m0); hi1 = fmax(hi1, v.m1); hi2 = fmax(hi2, v.m2);
	};
	// Two rows of this thread per batch (ra, ra + 256): both descriptions, then both records, then the stores — vector memory
	// operations retire in issue order (vmcnt counts loads and stores together), so no load is issued behind a store of its
	// own batch. (Four rows per batch: the kernel spilled 192 bytes per lane at its 128 registers.)
	double kmean[4][3];
#pragma unroll
	for (int q = 0; q < 4; q++) { kmean[q][0] = 0; kmean[q][1] = 0; kmean[q][2] = 0; }
	const bool anybirth = npredicted > nprior;
	auto gather2 = [&](int ra, double* ka, double* kb) {
		const int rb = ra + 256;
		const int ca = min(ra, cut - 1), cb = min(rb, cut - 1);
		RowSrc oa, ob;
		if (rk_kw && rk_ki) {   // (workgroup-uniform)
			oa = resolve_ranked(ca); ob = resolve_ranked(cb);
		}
		else {
			const int sa = rk_kw ? (int) rk_ks[order[ca]] : (int) (order[ca] & smask), sb = rk_kw ? (int) rk_ks[order[cb]] : (int) (order[cb] & smask);
			const double wa = a.emit_w[eb + sa], wb = a.emit_w[eb + sb];
			const int ia = a.emit_idx[eb + sa], ib = a.emit_idx[eb + sb];
			oa = resolve_listed(sa, wa, ia); ob = resolve_listed(sb, wb, ib);
		}
		RowRec va = fetch(oa), vb = fetch(ob);
		if (anybirth) { va = patch_birth(oa, va); vb = patch_birth(ob, vb); }   // (workgroup-uniform)
		if (ra < cut) { stage(ra, oa, va); if (ka) { ka[0] = va.m0; ka[1] = va.m1; ka[2] = va.m2; } }
		if (rb < cut) { stage(rb, ob, vb); if (kb) { kb[0] = vb.m0; kb[1] = vb.m1; kb[2] = vb.m2; } }
	};
	if (cut > 0) {
		gather2(tid, kmean[0], kmean[1]);          // (the means of the first four rows — all rows, up to 1024 kept entries —
		if (cut > 512) gather2(tid + 512, kmean[2], kmean[3]);   //  stay in registers for the grid below)
		for (int ra = tid + 1024; ra < cut; ra += 512) gather2(ra, nullptr, nullptr);
	}
	// the mean of slab row r: the 32 bytes at its head, one sector
	auto slab_mean = [&](int r, double& x0, double& x1, double& x2) {
		const double2* row = (const double2*) (srec + (size_t) r * PRUNE_ROW);
		const double2 u0 = row[0], u1 = row[1];
		x0 = u0.y; x1 = u1.x; x2 = u1.y;
	};
	{
		double red7[7] = {-lo0, -lo1, -lo2, hi0, hi1, hi2, rmx};   // all as maxima
#pragma unroll
		for (int q = 0; q < 7; q++) {
#pragma unroll
			for (int o = 32; o > 0; o >>= 1) red7[q] = fmax(red7[q], __shfl_xor(red7[q], o, 64));
		}
		if (lane == 0) {
#pragma unroll
			for (int q = 0; q < 7; q++) bred[wv * 7 + q] = red7[q];
		}
	}
	__syncthreads();   // the sort words are dead from here on: nbr / cand / owner / the cell lists take their place
	PHD_STAMP(2);

	// ---- B. all closeness tests close_i(k), k > i. The rows are binned in a uniform grid of cell size
	// 2 * (largest radius): the ball of a row then meets at most 2 x 2 x 2 cells, whose members are the only
	// candidates. Rows whose P_i is not positive definite, or whose radius exceeds the cell bound, test every
	// later row.
	{
		double red7[7];
#pragma unroll
		for (int q = 0; q < 7; q++) red7[q] = fmax(fmax(bred[q], bred[7 + q]), fmax(bred[14 + q], bred[21 + q]));
		const double mn0 = -red7[0], mn1 = -red7[1], mn2 = -red7[2];
		const double ext = fmax(red7[3] - mn0, fmax(red7[4] - mn1, red7[5] - mn2));
		double cell = fmax(2.02 * red7[6], ext / 60.0);   // <= 61 cells per axis
		if (!(cell > 0)) cell = 1.0;
		const double icell = 1.0 / cell, rcap = 0.5 * cell / 1.005;
		// float32 copies of the means relative to the box corner are off by at most 2^-24 ext per coordinate, so a
		// float32 distance is within ferr of the true one
		const double ferr = 4e-7 * ext;
		// Hash of a grid cell. The low three bits are the parities of the cell coordinates, so the 2 x 2 x 2 cells a
		// ball meets always fall into 8 different buckets (no bucket is walked twice); the rest mixes the halved
		// coordinates. Cells outside the box hash like any other: whatever they collide with fails the distance test.
		constexpr unsigned int HA = 73856093u, HB = 19349663u, HC = 83492791u;
		auto bucket = [](int cx, int cy, int cz) {
			const unsigned int h = (unsigned int) (cx >> 1) * HA ^ (unsigned int) (cy >> 1) * HB ^ (unsigned int) (cz >> 1) * HC;
			return (int) (((h & (PRUNE_NB / 8 - 1)) << 3) | (cx & 1) | ((cy & 1) << 1) | ((cz & 1) << 2));
		};
		for (int t = tid; t < (PRUNE_NB + 2) / 2 + 1; t += 256) cw[t] = 0;
		__syncthreads();
		// counting sort of the rows by bucket (16-bit counters: cut <= 65535)
		// (a row's mean: in this thread's registers since the gather — rows tid + 256 q, q < 4 — or, beyond 1024 kept
		// entries, back from its slab row)
		auto count_row = [&](int r, double x0, double x1, double x2) {
			const double s0 = x0 - mn0, s1 = x1 - mn1, s2 = x2 - mn2;
			const int b = bucket((int) (s0 * icell), (int) (s1 * icell), (int) (s2 * icell));
			owner[r] = b;
			atomicAdd(&cw[b >> 1], 1u << (16 * (b & 1)));
		};
#pragma unroll
		for (int q = 0; q < 4; q++) {
			if (tid + 256 * q < cut) count_row(tid + 256 * q, kmean[q][0], kmean[q][1], kmean[q][2]);
		}
		for (int r = tid + 1024; r < cut; r += 256) {
			double x0, x1, x2;
			slab_mean(r, x0, x1, x2);
			count_row(r, x0, x1, x2);
		}
		__syncthreads();
		{   // inclusive prefix over the bucket counts: cstart[b] = end of bucket b; the fill below counts it down to its start
			constexpr int WT = PRUNE_NB / 2 / 256;   // words per thread
			unsigned int wd[WT];
			int tot = 0;
#pragma unroll
			for (int u = 0; u < WT; u++) { wd[u] = cw[WT * tid + u]; tot += (int) (wd[u] & 0xffff) + (int) (wd[u] >> 16); }
			int incl = tot;
#pragma unroll
			for (int o = 1; o < 64; o <<= 1) {
				int y = __shfl_up(incl, o, 64);
				if (lane >= o) incl += y;
			}
			if (lane == 63) scan[wv] = incl;
			__syncthreads();
			int run = incl - tot;
			for (int q = 0; q < wv; q++) run += scan[q];
#pragma unroll
			for (int u = 0; u < WT; u++) {
				const int e0 = run + (int) (wd[u] & 0xffff), e1 = e0 + (int) (wd[u] >> 16);
				cw[WT * tid + u] = (unsigned int) e0 | ((unsigned int) e1 << 16);
				run = e1;
			}
			if (tid == 0) cw[PRUNE_NB / 2] = (unsigned int) cut;   // cstart[NB]: the end of the last bucket
		}
		__syncthreads();
		auto fill_row = [&](int r, double x0, double x1, double x2) {
			const double s0 = x0 - mn0, s1 = x1 - mn1, s2 = x2 - mn2;
			const int b = owner[r], sh = 16 * (b & 1);
			const int pos = (int) ((atomicSub(&cw[b >> 1], 1u << sh) >> sh) & 0xffff) - 1;
			cand[pos] = make_float4((float) s0, (float) s1, (float) s2, __int_as_float(r));
			rowpos[r] = (unsigned short) pos;
			owner[r] = -1;
		};
#pragma unroll
		for (int q = 0; q < 4; q++) {
			if (tid + 256 * q < cut) fill_row(tid + 256 * q, kmean[q][0], kmean[q][1], kmean[q][2]);
		}
		for (int r = tid + 1024; r < cut; r += 256) {
			double x0, x1, x2;
			slab_mean(r, x0, x1, x2);
			fill_row(r, x0, x1, x2);
		}
		__syncthreads();
		PHD_STAMP(7);

		// The rows are handed to the lanes in the order of their candidate counts (the rows of the 8 buckets their ball
		// meets; cells near the sensor hold an order of magnitude more rows than the rest): a wave then walks as long as
		// its rows need on average, not as long as the one unlucky row in it. Counting sort by min(count, 63); 63 also
		// stands for the rows that must test every later row.
		// (the row's own float32 coordinates, from its record in `cand`: 2^-24 of the extent away from the true ones, far
		// inside the margin rcap leaves between a ball and the 2 x 2 x 2 cells around it — no global load in this part)
		auto ranges = [&](const float4& me, unsigned int* qb) {   // the 8 bucket ranges of a row as start | length << 16; returns the total
			const int bx = (int) floor((double) me.x * icell - 0.5), by = (int) floor((double) me.y * icell - 0.5),
			          bz = (int) floor((double) me.z * icell - 0.5);
			const unsigned int xa0 = (unsigned int) (bx >> 1) * HA, xa1 = (unsigned int) ((bx + 1) >> 1) * HA;
			const unsigned int yb0 = (unsigned int) (by >> 1) * HB, yb1 = (unsigned int) ((by + 1) >> 1) * HB;
			const unsigned int zc0 = (unsigned int) (bz >> 1) * HC, zc1 = (unsigned int) ((bz + 1) >> 1) * HC;
			const int par0 = (bx & 1) | ((by & 1) << 1) | ((bz & 1) << 2);
			int n = 0;
#pragma unroll
			for (int c = 0; c < 8; c++) {
				const unsigned int h = ((c & 1) ? xa1 : xa0) ^ ((c & 2) ? yb1 : yb0) ^ ((c & 4) ? zc1 : zc0);
				const int bk = (int) ((h & (PRUNE_NB / 8 - 1)) << 3) | (par0 ^ c);   // neighbours flip the parity bits
				const int s0 = cstart[bk], n0 = cstart[bk + 1] - s0;
				qb[c] = (unsigned int) s0 | ((unsigned int) n0 << 16);
				n += n0;
			}
			return n;
		};
		int* hist = scan;   // [64]
		if (tid < 64) hist[tid] = 0;
		__syncthreads();
		for (int i = tid; i < cut; i += 256) {
			int key = 63;
			if ((double) rad2[i] <= rcap * rcap) {
				unsigned int qb[8];
				key = min(ranges(cand[rowpos[i]], qb), 62);
			}
			owner[i] = key;
			atomicAdd(&hist[key], 1);
		}
		__syncthreads();
		if (tid < 64) {   // hist[key] = first position of the rows with that key, lightest first: thread t takes the positions
			// t, t + 256, ..., so the waves that get a last, partial round of the heaviest rows started with the lightest
			const int h = hist[tid];
			int incl = h;
#pragma unroll
			for (int o = 1; o < 64; o <<= 1) {
				int y = __shfl_up(incl, o, 64);
				if (lane >= o) incl += y;
			}
			hist[tid] = incl - h;
		}
		__syncthreads();
		for (int i = tid; i < cut; i += 256) {
			ord[atomicAdd(&hist[owner[i]], 1)] = (unsigned short) i;
			owner[i] = -1;
		}
		__syncthreads();
		PHD_STAMP(8);
		// hist[key] now ends key's rows: the rows with more than PRUNE_HEAVY candidates (and those on the full scan) sit
		// at the positions from hstart on. One lane would walk such a row for a hundred trips while its wave waits, so
		// they are left to the loop after this one, where a wave spreads one row's candidates over its lanes.
		const int hstart = hist[PRUNE_HEAVY];
#ifdef PHD_STAMPS
		long long acc_setup = 0, acc_trips = 0, acc_drain = 0, tmark = clock64();
#endif

		for (int t = tid; t < hstart; t += 256) {
			const int i = ord[t];
			const float4 me = cand[rowpos[i]];
			const double bound = (double) rad2[i];
			int cnt = 0;
			unsigned int e[PRUNE_NBR];   // close rows found (statically indexed only)
#pragma unroll
			for (int q = 0; q < PRUNE_NBR; q++) e[q] = 0xffffu;
			// the row's own record (mean, P^-1) is needed by the exact test only, which few rows ever reach: loaded then
			bool have = false;
			double m0 = 0, m1 = 0, m2 = 0, Pi[6] = {0, 0, 0, 0, 0, 0};
			auto test = [&](int k) {
				if (!have) {
					double P[6], mm[3], wi, det;
					load_comp(srec + (size_t) i * PRUNE_ROW, wi, mm, P);
					inv_sym3(P, Pi, det);
					m0 = mm[0]; m1 = mm[1]; m2 = mm[2];
					have = true;
				}
				double k0, k1, k2;
				slab_mean(k, k0, k1, k2);
				double d0 = m0 - k0, d1 = m1 - k1, d2 = m2 - k2;
				double sq = d0 * d0 + d1 * d1 + d2 * d2;
				if (sq <= bound && quad_sym(Pi, d0, d1, d2) < merge_thr2) {   // Gaussian.SquareMahalanobis(b.Mean) < threshold^2
					// keep the 7 smallest row numbers, ascending (insertion through a fixed network)
					unsigned int v = (unsigned int) k;
#pragma unroll
					for (int q = 0; q < PRUNE_NBR; q++) {
						unsigned int lo_ = min(e[q], v), hi_ = max(e[q], v);
						e[q] = lo_; v = hi_;
					}
					cnt++;
				}
			};
			if (bound <= rcap * rcap) {
				const float fx = me.x, fy = me.y, fz = me.z;
				const double rr = sqrt(bound) + ferr;
				const float thr = (float) (rr * rr * (1.0 + 1e-5));
				unsigned long long pend0 = 0, pend1 = 0;   // up to 8 queued rows as 16-bit fields
				int npend = 0;
				// the queued rows' exact tests, four at a time: their means are fetched together (one trip to the slab for the
				// four, not one each, one after the other)
				auto drain = [&]() {
					if (npend > 0 && !have) {
						double P[6], mm[3], wi, det;
						load_comp(srec + (size_t) i * PRUNE_ROW, wi, mm, P);
						inv_sym3(P, Pi, det);
						m0 = mm[0]; m1 = mm[1]; m2 = mm[2];
						have = true;
					}
					for (int b0 = 0; b0 < npend; b0 += 4) {
						int kk[4];
						double kx[4], ky[4], kz[4];
#pragma unroll
						for (int u = 0; u < 4; u++) {
							const int f = b0 + u;
							kk[u] = (f < npend) ? (int) (((f < 4) ? (pend0 >> (16 * f)) : (pend1 >> (16 * (f - 4)))) & 0xffff) : i;
							slab_mean(kk[u], kx[u], ky[u], kz[u]);   // (an idle slot reads the row's own mean: a line it has)
						}
#pragma unroll
						for (int u = 0; u < 4; u++) {
							const double d0 = m0 - kx[u], d1 = m1 - ky[u], d2 = m2 - kz[u];
							const double sq = d0 * d0 + d1 * d1 + d2 * d2;
							if (b0 + u < npend && sq <= bound && quad_sym(Pi, d0, d1, d2) < merge_thr2) {
								unsigned int v = (unsigned int) kk[u];
#pragma unroll
								for (int q = 0; q < PRUNE_NBR; q++) {
									unsigned int lo_ = min(e[q], v), hi_ = max(e[q], v);
									e[q] = lo_; v = hi_;
								}
								cnt++;
							}
						}
					}
					npend = 0; pend0 = 0; pend1 = 0;
				};
				unsigned int qb[8];
				const int n = ranges(me, qb);
				// one candidate per trip, the 8 ranges one after the other. The exact test is rare per lane but not per
				// wave: a candidate that passes the float32 distance test is queued and tested later, so that the waves do
				// not run the FP64 path at every candidate
				// the non-empty ranges first, in their order (a stable compaction through static indices): the walk below then
				// moves to the next range with one predicated step instead of a loop over possibly empty ones
				unsigned int nz[8];
				{
					int before[8], run = 0;
#pragma unroll
					for (int u = 0; u < 8; u++) { before[u] = run; run += (qb[u] >> 16) ? 1 : 0; }
#pragma unroll
					for (int j = 0; j < 8; j++) {
						unsigned int v = 0;
#pragma unroll
						for (int u = j; u < 8; u++) v = ((qb[u] >> 16) && before[u] == j) ? qb[u] : v;
						nz[j] = v;
					}
				}
#ifdef PHD_STAMPS
				{ long long now = clock64(); acc_setup += now - tmark; tmark = now; }
#endif
				unsigned int cur = 0, left = 0;
				int c = 0;
				// (four candidates per trip, their four LDS reads in flight together, was measured: the walk took 37 k cycles
				// against 19 k — the bookkeeping of four cursors with their divergent range changes costs more than the latency)
				for (int f = 0; f < n; f++) {
#ifdef PHD_STAMP_COUNTERS
					if (a.stamps && a.stamp_kernel == 2 && (int) (__ffsll((long long) ballot64(1)) - 1) == lane) atomicAdd(&a.stamps[(size_t) p * 16 + 15], 1.0);   // wave-level trips
#endif
					if (left == 0) {
						unsigned int q = nz[0];
#pragma unroll
						for (int u = 1; u < 8; u++) q = (c == u) ? nz[u] : q;
						c++;
						cur = q & 0xffff; left = q >> 16;
					}
					const float4 cd = cand[cur];
					cur++; left--;
					const int k = __float_as_int(cd.w);
					const float e0 = fx - cd.x, e1 = fy - cd.y, e2 = fz - cd.z;
					if (k > i && e0 * e0 + e1 * e1 + e2 * e2 <= thr) {
						if (npend < 4) pend0 |= (unsigned long long) k << (16 * npend);
						else           pend1 |= (unsigned long long) k << (16 * (npend - 4));
						npend++;
						if (npend > 7) drain();
					}
				}
#ifdef PHD_STAMPS
				{ long long now = clock64(); acc_trips += now - tmark; tmark = now; }
#endif
				drain();
#ifdef PHD_STAMPS
				{ long long now = clock64(); acc_drain += now - tmark; tmark = now; }
#endif
			}
			else {
				for (int k = i + 1; k < cut; k++) test(k);
			}
			unsigned long long lo = (unsigned long long) min(cnt, 0xffff), hi = 0;
#pragma unroll
			for (int q = 0; q < PRUNE_NBR; q++) {
				if (q < cnt) {
					if (q < 3) lo |= (unsigned long long) e[q] << (16 * (q + 1));
					else hi |= (unsigned long long) e[q] << (16 * (q - 3));
				}
			}
			nbr[2 * i]     = lo;
			nbr[2 * i + 1] = hi;
#ifdef PHD_STAMP_COUNTERS   // (slow: contended atomics; counts only, never together with timing)
			if (a.stamps && a.stamp_kernel == 2) {
				atomicAdd(&a.stamps[(size_t) p * 16 + 14], (double) cnt);                               // close pairs
				unsigned int qd[8];
				atomicAdd(&a.stamps[(size_t) p * 16 + 13], (double) ((bound <= rcap * rcap) ? ranges(me, qd) : 0));     // candidates
			}
#endif
		}

#if defined(PHD_STAMPS) && !defined(PHD_STAMP_COUNTERS)   // (the counters' build uses the same slots for its counts)
		if (tid == 0 && a.stamps && a.stamp_kernel == 2) { a.stamps[(size_t) p * 16 + 12] = (double) acc_setup; a.stamps[(size_t) p * 16 + 13] = (double) acc_trips; a.stamps[(size_t) p * 16 + 14] = (double) acc_drain; }
#endif
		PHD_STAMP(9);
		for (int h = hstart + wv; h < cut; h += 4) {   // (wave-uniform) one crowded row per wave at a time, candidate per lane
			const int i = ord[h];
			const float4 me = cand[rowpos[i]];
			const double bound = (double) rad2[i];
			int cnt = 0;
			unsigned int e[PRUNE_NBR];   // the same in every lane
#pragma unroll
			for (int q = 0; q < PRUNE_NBR; q++) e[q] = 0xffffu;
			bool have = false;
			double m0 = 0, m1 = 0, m2 = 0, Pi[6] = {0, 0, 0, 0, 0, 0};
			auto exact = [&](int k) {
				if (!have) {
					double P[6], mm[3], wi, det;
					load_comp(srec + (size_t) i * PRUNE_ROW, wi, mm, P);
					inv_sym3(P, Pi, det);
					m0 = mm[0]; m1 = mm[1]; m2 = mm[2];
					have = true;
				}
				double k0, k1, k2;
				slab_mean(k, k0, k1, k2);
				const double d0 = m0 - k0, d1 = m1 - k1, d2 = m2 - k2;
				return d0 * d0 + d1 * d1 + d2 * d2 <= bound && quad_sym(Pi, d0, d1, d2) < merge_thr2;
			};
			auto collect = [&](bool close, int k) {   // the close rows found by the lanes, into the sorted list all lanes keep
				unsigned long long bal = ballot64(close);
				while (bal) {
					const int l = __ffsll((long long) bal) - 1;
					bal &= bal - 1;
					unsigned int v = (unsigned int) __shfl(k, l, 64);
#pragma unroll
					for (int q = 0; q < PRUNE_NBR; q++) {
						unsigned int lo_ = min(e[q], v), hi_ = max(e[q], v);
						e[q] = lo_; v = hi_;
					}
					cnt++;
				}
			};
			if (bound <= rcap * rcap) {
				const double rr = sqrt(bound) + ferr;
				const float thr = (float) (rr * rr * (1.0 + 1e-5));
				unsigned int qb[8];
				const int n = ranges(me, qb);
				for (int f0 = 0; f0 < n; f0 += 64) {
					const int f = f0 + lane;
					int rem = f;
					unsigned int cur = 0;
					bool found = false;
#pragma unroll
					for (int c = 0; c < 8; c++) {
						const int len = (int) (qb[c] >> 16);
						if (!found && rem < len) { cur = (qb[c] & 0xffff) + (unsigned int) rem; found = true; }
						else if (!found) rem -= len;
					}
					bool close = false;
					int k = 0;
					if (found) {
						const float4 cd = cand[cur];
						k = __float_as_int(cd.w);
						const float e0 = me.x - cd.x, e1 = me.y - cd.y, e2 = me.z - cd.z;
						if (k > i && e0 * e0 + e1 * e1 + e2 * e2 <= thr) close = exact(k);
					}
					collect(close, k);
				}
			}
			else {
				for (int k0 = i + 1; k0 < cut; k0 += 64) {
					const int k = k0 + lane;
					collect(k < cut && exact(k), k);
				}
			}
			if (lane == 0) {
				unsigned long long lo = (unsigned long long) min(cnt, 0xffff), hi = 0;
#pragma unroll
				for (int q = 0; q < PRUNE_NBR; q++) {
					if (q < cnt) {
						if (q < 3) lo |= (unsigned long long) e[q] << (16 * (q + 1));
						else hi |= (unsigned long long) e[q] << (16 * (q - 3));
					}
				}
				nbr[2 * i]     = lo;
				nbr[2 * i + 1] = hi;
			}
		}
	}
	PHD_STAMP(10);
	__syncthreads();

	PHD_STAMP(3);
	// ---- C. who survives: sequential in rank order, one wave, absorbed bits in registers
	if (wv == 0) {
		unsigned int absorbed = 0;                  // bit s of lane l <-> row s*64 + l
		const int nslots = (cut + 63) >> 6;
		for (int base = 0; base < cut; base += 64) {
			const int s0 = base >> 6;
			// this lane holds the record of row base + lane
			unsigned long long vlo = 0, vhi = 0;
			if (base + lane < cut) { vlo = nbr[2 * (base + lane)]; vhi = nbr[2 * (base + lane) + 1]; }
			const int lolo = (int) (unsigned int) vlo, lohi = (int) (unsigned int) (vlo >> 32);
			const int hilo = (int) (unsigned int) vhi, hihi = (int) (unsigned int) (vhi >> 32);
			// only rows that have close later rows can change anything: walk those, in order
			unsigned long long todo = ballot64((lolo & 0xffff) != 0);
			while (todo) {
				const int l = __ffsll((long long) todo) - 1;
				todo &= todo - 1;
				const int i = base + l;
				unsigned int om = (unsigned int) __builtin_amdgcn_readlane((int) absorbed, l);
				if ((om >> s0) & 1u) continue;
				const unsigned int w0 = (unsigned int) __builtin_amdgcn_readlane(lolo, l);
				const int cnt = (int) (w0 & 0xffff);
				if (cnt <= PRUNE_NBR) {
					const unsigned long long lo = ((unsigned long long) (unsigned int) __builtin_amdgcn_readlane(lohi, l) << 32) | w0;
					const unsigned long long hi = ((unsigned long long) (unsigned int) __builtin_amdgcn_readlane(hihi, l) << 32) |
					                              (unsigned int) __builtin_amdgcn_readlane(hilo, l);
					for (int c = 0; c < cnt; c++) {
						int k = (int) (((c < 3) ? (lo >> (16 * (c + 1))) : (hi >> (16 * (c - 3)))) & 0xffff);
						unsigned int km = (unsigned int) __builtin_amdgcn_readlane((int) absorbed, k & 63);
						if (!((km >> (k >> 6)) & 1u)) {
							if (lane == (k & 63)) absorbed |= 1u << (k >> 6);
							if (lane == 0) owner[k] = i;
						}
					}
				}
				else {
					// more close rows than the list holds: re-test row i against every later row, 64 at a time
					double P[6], Pi[6], mm[3], wi, det;
					load_comp(srec + (size_t) i * PRUNE_ROW, wi, mm, P);
					inv_sym3(P, Pi, det);
					const double m0 = mm[0], m1 = mm[1], m2 = mm[2];
					for (int s = i >> 6; s < nslots; s++) {
						int k = s * 64 + lane;
						if (k > i && k < cut && !((absorbed >> s) & 1u)) {
							double k0, k1, k2;
							slab_mean(k, k0, k1, k2);
							if (quad_sym(Pi, m0 - k0, m1 - k1, m2 - k2) < merge_thr2) {
								absorbed |= 1u << s;
								owner[k] = i;
							}
						}
					}
				}
			}
		}
		absb[lane] = (int) absorbed;
	}
	__syncthreads();

	PHD_STAMP(4);
	// ---- D. output position of every survivor (exclusive scan of the survivor flags) and the merges.
	// For the reweight (k_alpha_density): a survivor that absorbed nothing and is the misdetection copy of predicted
	// component c (canonical index c < np, PHDNavigator.cs:837-840) IS that component with another weight — Merge
	// (Gaussian.cs:329-346) of a single Gaussian changes its moments only by rounding, which is checked here —, so its
	// density at a landmark is the predicted component's times wcopy[c] / w_c. wcopy[c] = that weight (0: none such),
	// cover[pos] = 1 for the survivors accounted for this way.
	const int npred = bank_of(a, SEL_IN).count[p] + a.born_count[p];
	double* wcopy = a.wcopy + (size_t) p * (a.cap + a.Mcap);
	for (int c = tid; c < npred; c += 256) wcopy[c] = 0.0;
	__syncthreads();
	// positions first, for all rows at once (one barrier): the survivors of every chunk of 256 rows and wave are counted into
	// scan[chunk][wave]; a survivor's position is the survivors of the chunks before its own, of the waves before its own in
	// the chunk, and of the lanes before its own in the wave
	const int nchunks = (cut + 255) >> 8;   // (at most 14: MaxQuantity is bounded by the LDS, phd_create)
	auto survives = [&](int i) { return i < cut && !(((unsigned int) absb[i & 63] >> (i >> 6)) & 1u); };
	for (int q = 0; q < nchunks; q++) {
		const unsigned long long bal = ballot64(survives(q * 256 + tid));
		if (lane == 0) scan[q * 4 + wv] = __popcll(bal);
	}
	__syncthreads();
	int nsurv_before = 0;
	// A thread's rows are tid + 256 q, one after the other; the slab record of the NEXT one is requested before the results of
	// this one are stored (memory operations retire in issue order: behind the stores the next row's loads would wait for the
	// stores' acknowledgements as well).
	double nw = 0, nm[3] = {0, 0, 0}, nP[6] = {0, 0, 0, 0, 0, 0}, ncidxd = 0;
	auto request = [&](int q) {   // (no branch: a row beyond the cut, or an absorbed one, reads a line its wave reads anyway)
		const int i = min(q * 256 + tid, max(cut - 1, 0));
		const double2* row = (const double2*) (srec + (size_t) i * PRUNE_ROW);
		const double2 r0 = row[0], r1 = row[1], r2 = row[2], r3 = row[3], r4 = row[4], r5 = row[5];
		nw = r0.x; nm[0] = r0.y; nm[1] = r1.x; nm[2] = r1.y;
		nP[0] = r2.x; nP[1] = r2.y; nP[2] = r3.x; nP[3] = r3.y; nP[4] = r4.x; nP[5] = r4.y;
		ncidxd = r5.x;
	};
	if (cut > 0) request(0);
	for (int q = 0; q < nchunks; q++) {
		const int  i = q * 256 + tid;
		const bool surv = survives(i);
		const unsigned long long bal = ballot64(surv);
		int base = nsurv_before;
		for (int u = 0; u < wv; u++) base += scan[q * 4 + u];
		const int total = scan[q * 4] + scan[q * 4 + 1] + scan[q * 4 + 2] + scan[q * 4 + 3];
		// (this row's slab record: the component record and, in the sixth 16 bytes, its canonical index)
		const double w = nw, Pr[6] = {nP[0], nP[1], nP[2], nP[3], nP[4], nP[5]};
		const double m0 = nm[0], m1 = nm[1], m2 = nm[2];
		const int cidx = (int) ncidxd;
		if (q + 1 < nchunks) request(q + 1);   // (uniform)
		if (surv) {
			// the reference runs on the CLR (no fused multiply-add): Merge's raw moments must round as there, or a scene far
			// from the origin loses digits of the covariance to the difference (tests: test_reweight_far_from_the_origin)
PHD_REF_ARITH
			const int pos = base + __popcll(bal & lanemask_lt());
			// Gaussian.Merge (Gaussian.cs:329-346): raw moments, the candidate first, then its set in list order
			double W = 0.0 + w;
			double M0 = 0.0 + w * m0, M1 = 0.0 + w * m1, M2 = 0.0 + w * m2;
			double C0 = 0.0 + w * (Pr[0] + m0 * m0), C1 = 0.0 + w * (Pr[1] + m0 * m1), C2 = 0.0 + w * (Pr[2] + m0 * m2);
			double C3 = 0.0 + w * (Pr[3] + m1 * m1), C4 = 0.0 + w * (Pr[4] + m1 * m2), C5 = 0.0 + w * (Pr[5] + m2 * m2);
			int nabs = 0;
			auto absorb = [&](int k) {
PHD_REF_ARITH
				nabs++;
				double wk, mk[3], Pk[6];
				load_comp(srec + (size_t) k * PRUNE_ROW, wk, mk, Pk);
				const double k0 = mk[0], k1 = mk[1], k2 = mk[2];
				W += wk;
				M0 += wk * k0; M1 += wk * k1; M2 += wk * k2;
				C0 += wk * (Pk[0] + k0 * k0); C1 += wk * (Pk[1] + k0 * k1); C2 += wk * (Pk[2] + k0 * k2);
				C3 += wk * (Pk[3] + k1 * k1); C4 += wk * (Pk[4] + k1 * k2); C5 += wk * (Pk[5] + k2 * k2);
			};
			const unsigned long long lo = nbr[2 * i], hi = nbr[2 * i + 1];
			const int cnt = (int) (lo & 0xffff);
			if (cnt <= PRUNE_NBR) {
				for (int c = 0; c < cnt; c++) {
					int k = (int) (((c < 3) ? (lo >> (16 * (c + 1))) : (hi >> (16 * (c - 3)))) & 0xffff);
					if (owner[k] == i) absorb(k);
				}
			}
			else {
				for (int k = i + 1; k < cut; k++) {
					if (owner[k] == i) absorb(k);
				}
			}
			double ow, o0, o1, o2, oP[6];
			if (W < 1e-15) {   // Gaussian.cs:339-341
				ow = 0.0; o0 = m0; o1 = m1; o2 = m2;
				oP[0] = 1e12; oP[1] = 0; oP[2] = 0; oP[3] = 1e12; oP[4] = 0; oP[5] = 1e12;
			}
			else {
				ow = W; o0 = M0 / W; o1 = M1 / W; o2 = M2 / W;
				oP[0] = C0 / W - o0 * o0; oP[1] = C1 / W - o0 * o1; oP[2] = C2 / W - o0 * o2;
				oP[3] = C3 / W - o1 * o1; oP[4] = C4 / W - o1 * o2; oP[5] = C5 / W - o2 * o2;
			}
			const size_t ob = (size_t) p * a.cap + pos;
			{
				const double om[3] = {o0, o1, o2};
				store_comp(vout.rec + ob * MIX_REC, ow, om, oP);
			}
			a.outw[ob] = ow;   // (the weights once more as a plane: all that BestMapEstimate reads of most components)
			int covered = 0;
			if (nabs == 0 && cidx < npred && !(W < 1e-15)) {
				const double pscale = 1e-9 * fmax(fabs(Pr[0]), fmax(fabs(Pr[3]), fabs(Pr[5])));
				bool same = fabs(o0 - m0) <= 1e-9 * fabs(m0) && fabs(o1 - m1) <= 1e-9 * fabs(m1) && fabs(o2 - m2) <= 1e-9 * fabs(m2);
#pragma unroll
				for (int t = 0; t < 6; t++) same = same && fabs(oP[t] - Pr[t]) <= pscale;
				if (same) { wcopy[cidx] = ow; covered = 1; }
			}
			a.cover[ob] = covered;
		}
		nsurv_before += total;
	}
	PHD_STAMP(5);
	if (tid == 0) vout.count[p] = nsurv_before;
	PHD_STAMP_FLUSH(2, 12);
#if defined(PHD_STAMPS) && !defined(PHD_STAMP_COUNTERS)   // (the fused launch: cycles of the emit body in front of this one, slot 15)
	if (tid == 0 && a.stamps && a.stamp_kernel == 2 && tk0) a.stamps[(size_t) p * 16 + 15] = (double) (stamp_[0] - tk0);
#endif
}

#ifndef PHD_PRUNE_WAVES
#define PHD_PRUNE_WAVES 4
#endif
__global__ __launch_bounds__(256, PHD_PRUNE_WAVES) void k_prune_merge(const DevParams prm, const StepBufs a, int cutcap)
{
	extern __shared__ __align__(16) double smem[];
	prune_merge_body(prm, a, cutcap, smem);
}

// k_emit_finish and k_prune_merge as ONE launch (environment PHD_FUSE_EP=1): a particle's corrected list goes from the one
// body to the other behind a workgroup barrier, the launch boundary between two latency-bound kernels — a tail in which the
// machine drains and a ramp in which it fills — is paid once. Registers: the Kalman path compiled at this kernel's 128.
__global__ __launch_bounds__(256, PHD_PRUNE_WAVES) void k_emit_prune(const DevParams prm, const StepBufs a, int cutcap)
{
	extern __shared__ __align__(16) double smem[];
	PHD_TL_BEGIN;
	PHD_SET_PRIO(PHD_LAT_PRIO);
#ifdef PHD_STAMPS
	const long long tk0 = clock64();
#else
	const long long tk0 = 0;
#endif
	emit_finish_body<true>(prm, a, smem);
	__syncthreads();   // (workgroup scope: what this workgroup's waves stored is visible to its loads behind the barrier)
	prune_merge_body(prm, a, cutcap, smem, tk0);
	PHD_TL_END(2);
}
