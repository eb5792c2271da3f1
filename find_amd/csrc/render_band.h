// raster_band_kernel: the list-free K-nearest rule of round 5 (commits e7c5af6 / a265fe6), kept BESIDE raster_kernel.  It wins where
// tiles are ordinary -- 512^2 renders: 2.27 against 2.75 ms for 16 template feet x 4 views -- and loses where single depth slabs hold a thousand
// candidates per pixel (the poles of the latitude-longitude GT scans at 256^2), so find_render_fwd picks by image size ("band" switch of
// find_render_switches to force either).  Same (pixel, face) arithmetic, same K-nearest set, ties at the K-th depth by face index in place
// (no fix-up queue); the products of 1 - p are formed band by band, so alpha can differ from raster_kernel's in the last bits.
// Included by render.hip inside namespace find::render, behind the helpers both kernels share.
#pragma once

// ------------------------------------------------------------------------------------------------ 4. rasteriser
// Persistent waves, each a worker of its own: tile from the queues -> list from the pool, 64 faces at a time (the next 64 entries are on
// their way from memory while these are evaluated) -> every lane evaluates the staged records for its own pixel.
//
// The K-nearest rule WITHOUT per-pixel candidate lists (round 5; rounds 2-4 appended every candidate -- depth, 1 - p: 8 B -- to a list in
// scratch memory and searched it afterwards: 0.42 GB written and ~0.5 GB re-read per C3 launch, 1.7 GB at 512^2, a third to a half of the
// kernel's time).  The list is in depth-slab order: while batch i is evaluated every face still to come lies behind `front` (F_i, the lower
// edge of the next batch's first slab), so at the end of batch i the number of candidates in front of F_i, N_i, is FINAL.  A pixel keeps
// counts and PRODUCTS of (1 - p) in three bands -- in front of F_i, within one slab behind it, further back (never beyond two) -- which
// move up as the front moves.  N_i < K: those candidates all belong to the K nearest, their product is all that is needed of them.  The
// first i with N_i >= K ("crossing") fixes the band [F_{i-1}, F_i) that holds the K-th nearest: N_i == K -- alpha is the product in front
// of F_i, done; N_i > K -- the (K - N_{i-1})-th smallest depth of THAT BAND decides.  Only for those pixels a second sweep re-evaluates
// the few batches whose slabs can reach their band (face ids at hand), writes the band's candidates (depth, 1 - p, face: a tenth of what
// the lists held) to the lane's scratch, finds the rank by a radix search on the depth bits and resolves ties at the K-th depth by face
// index right there (PyTorch3D's K-buffer keeps the lower indices: rounds 2-4 needed a fix-up kernel with a queue for that).
// A pixel that never crosses has fewer than K candidates: alpha is the product of all of them.
struct RasterBandArgs {   // (what the rasteriser's loop needs and no more: the shading tail and its dozen pointers live in shade_kernel)
	float sil_blur_radius, sil_sigma;
	int sil_faces_per_pixel, image_h, image_w;
	const uint32_t* tb;
	const int32_t* zinfo;
	const int2* tinfo;
	int F, tiles_x, tiles_per_img;
	int64_t pool_cap;
	float* mask;
	int32_t* p2f_ws;      // null: no colour pass
	float* frag_ws;
	int32_t* flags;
	const int32_t* qn;
	float* zthr;
	float* alpha_ws;
	int32_t* tie_face;
	float* scratch;
	int ablate;
};


template <bool want_sil, bool want_rgb>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void raster_band_kernel(const RasterBandArgs a, const FaceRec* __restrict__ recs, const uint32_t* __restrict__ pool_all,
													  const int32_t* __restrict__ order) {
	// the batch in flight, as PAIRS of faces: 32 blocks of 64 floats per wave, field j of the pair's faces at [2 j], [2 j + 1] (eval_pair);
	// the second sweep stages flat records in it, the radix search's counters live on top of it afterwards
	__shared__ __attribute__((aligned(16))) float rec[4][32 * PAIR_STRIDE];   // (>= 64 * REC_DW: the second sweep's flat records)
	static_assert(32 * PAIR_STRIDE >= 64 * REC_DW, "the flat records of the second sweep share the pair blocks' space");
	__shared__ unsigned band_n[4][192];   // second sweep: per pixel of the tile the band candidates delivered so far, bits of their smallest / largest depth
	__shared__ unsigned short item_q[4][128];   // second sweep: queue of (pixel, face) pairs waiting for evaluation: pixel's lane << 8 | record

	const int H = a.image_h, W = a.image_w;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const float blur = a.sil_blur_radius;
	const float inv_sigma = 1.0f / a.sil_sigma;
	const int K = a.sil_faces_per_pixel;
	const bool early = !(a.ablate & 8);
	const int n_order = a.qn[62];
	// band candidates of the second sweep: entry j of lane l at [j * 64 + l] of the wave's three arrays (depth, 1 - p, face)
	float* const wz0 = a.scratch + ((int64_t)blockIdx.x * 4 + wave) * (3 * (int64_t)KN_CAP * 64);
	float* const wq0 = wz0 + KN_CAP * 64;
	int* const wf0 = reinterpret_cast<int*>(wq0 + KN_CAP * 64);
	float* const wz = wz0 + lane;   // this lane's column
	float* const wq = wq0 + lane;
	int* const wf = wf0 + lane;

	// (Measured and dropped: every wave's first tile by its own number instead of from the counter -- the 4096 first requests queue up
	// for ~48 us on that one address -- made the kernel 0.4 ms SLOWER: the staggered start spreads the longest lists, which all sit at
	// the head of the order, over time and over the CUs.)
	for (;;) {
		int t_q = 0;
		if (lane == 0) t_q = atomicAdd(&a.flags[3], 1);
		t_q = __builtin_amdgcn_readfirstlane(t_q);
		if (t_q >= n_order) break;
		const int t_id = __builtin_amdgcn_readfirstlane(order[t_q]);
		const int img = t_id / a.tiles_per_img, tile = t_id - img * a.tiles_per_img;
		const int tile_x = tile % a.tiles_x, tile_y = tile / a.tiles_x;
		const int xi = tile_x * T8 + (lane & 7), yi = tile_y * T8 + (lane >> 3);
		const bool in_img = xi < W && yi < H;
		const float px = 1.0f - (2.0f * xi + 1.0f) / (float)W;
		const float py = 1.0f - (2.0f * yi + 1.0f) / (float)H;
		int2 ti = a.tinfo[t_id];
		ti.x = __builtin_amdgcn_readfirstlane(ti.x); ti.y = __builtin_amdgcn_readfirstlane(ti.y);
		const bool binned = ti.y >= 0;
		const int n_list = binned ? (int)((uint32_t)ti.y & ~LIST_UNSORTED) : a.F;
		const bool sorted = binned && !((uint32_t)ti.y & LIST_UNSORTED);
		float zlo = 0.f, sw = 1.f;
		slab_layout(a.zinfo + img * 8, &zlo, &sw);
		const uint32_t* tbp = a.tb + (int64_t)img * a.F;
		const FaceRec* rp_img = recs + (int64_t)img * a.F;
		const uint32_t* lp = pool_all + (int64_t)img * a.pool_cap + ti.x;

		// silhouette state of this lane's pixel: candidates (count, product of 1 - p) in front of `front`, within one slab behind it, further back
		int c_lt = 0, c_a = 0, c_b = 0;
		float a_lt = 1.0f, a_a = 1.0f, a_b = 1.0f;
		// the last batch end at which fewer than K lay in front of the front (N_{i-1}, their product, F_{i-1} and its slab) ...
		int n_prev = 0, s_prev = -2;
		float a_prev = 1.0f, f_prev = 0.0f;
		// ... and the first at which K or more do: the crossing
		bool crossed = false;
		int n_at = 0, s_at = 0;
		float a_at = 1.0f, f_at = INFINITY;
		int n_eval = 0;
		float bz = INFINITY, bd = 0.f, bw0 = 0.f, bw1 = 0.f, bw2 = 0.f;
		int bf = -1;
		float front = 0.0f, front1 = 0.0f;   // every face not yet evaluated has its fragments behind `front`; front1: one slab further
		int s_front = -2;                   // slab whose lower edge `front` is

		// entry of this lane in batch b: slab << 24 | face index (NONE: no face); *slab0: the slab of the batch's first entry
		constexpr uint32_t NONE = 0xFFFFFFFFu;
		auto entry = [&](int b, int* slab0) __attribute__((always_inline)) -> uint32_t {
			const int i = b * 64 + lane;
			uint32_t e = NONE, e0 = 0u;
			if (binned) {
				if (i < n_list) e = lp[i];
				if (b * 64 < n_list) e0 = lp[b * 64];
			} else if (i < n_list && tile_hit(tbp[i], tile_x, tile_y)) {
				e = (uint32_t)i;   // no room in the pool for this tile's list: every face of the image, tested here (slab 0: no front to back order)
			}
			*slab0 = (int)(e0 >> 24);
			return e;
		};
		// stage a batch: the faces present are packed to the front (the no-room path tests every face of the image: most lanes hold none)
		// and every lane scatters its face's 32 dwords into its half of a pair block; then every lane reads all the blocks.  Returns the count.
		auto stage = [&](uint32_t e) __attribute__((always_inline)) -> int {
			const unsigned long long m_have = __ballot(e != NONE);
			if (e != NONE) {
				const int pos = (int)__popcll(m_have & ((1ull << lane) - 1ull));
				const float4* src = reinterpret_cast<const float4*>(rp_img + (e & FACE_MASK));
				float4 q[REC_F4];
#pragma unroll
				for (int k = 0; k < REC_F4; ++k) q[k] = src[k];
				q[6].w = __int_as_float((int)(e & FACE_MASK));   // (the record's own face index, as written by face_setup_kernel)
				float* d = &rec[wave][(pos >> 1) * PAIR_STRIDE + (pos & 1)];
#pragma unroll
				for (int k = 0; k < REC_F4; ++k) { d[8 * k] = q[k].x; d[8 * k + 2] = q[k].y; d[8 * k + 4] = q[k].z; d[8 * k + 6] = q[k].w; }
			}
			wave_lds_sync();
			return (int)__popcll(m_have);
		};
		// RASTER_CHECKPOINT: the count in front of the CURRENT front is final whenever this runs (before the front moves, at a batch end): the
		// first time it reaches K is the crossing.  (A pixel that goes on for its colour after the crossing keeps counting: what the crossing
		// recorded is frozen.)  RASTER_ADVANCE(s_new): what is still to come lies behind the lower edge of slab s_new (the slab of the first
		// face not yet evaluated; N_SLABS + 1: nothing is).  The candidates seen so far come from faces of slabs up to that one, so none lies
		// two slabs behind the old edge: when the edge moves by one slab, those within a slab of the old one are now in front, the rest are
		// within a slab of the new one; by more, all are in front.
		// (Macros, and selects on VALUES: as lambdas capturing by reference -- and with `if (now) { n_at = ...; }` -- the compiler formed
		// stores through SELECTED ADDRESSES and kept this state in scratch memory, in the innermost loop.)
#define RASTER_CHECKPOINT()                                                                                                                       \
		if (want_sil) {                                                                                                                           \
			const bool now_ = !crossed & (c_lt >= K);                                                                                             \
			n_at = now_ ? c_lt : n_at; a_at = now_ ? a_lt : a_at; f_at = now_ ? front : f_at; s_at = now_ ? s_front : s_at;                      \
			crossed |= now_;                                                                                                                      \
			n_prev = crossed ? n_prev : c_lt; a_prev = crossed ? a_prev : a_lt; f_prev = crossed ? f_prev : front; s_prev = crossed ? s_prev : s_front; \
		}
#define RASTER_ADVANCE(s_new_)                                                                                                                    \
		{                                                                                                                                         \
			RASTER_CHECKPOINT();                                                                                                                  \
			const int sn_ = (s_new_);                                                                                                             \
			const bool one_ = sn_ == s_front + 1, end_ = sn_ > N_SLABS;                                                                          \
			const int ca_ = c_a, cb_ = c_b;                                                                                                       \
			const float aa_ = a_a, ab_ = a_b;                                                                                                     \
			c_lt += one_ ? ca_ : ca_ + cb_; a_lt *= one_ ? aa_ : aa_ * ab_;                                                                       \
			c_a = one_ ? cb_ : 0; a_a = one_ ? ab_ : 1.0f;                                                                                        \
			c_b = 0; a_b = 1.0f;                                                                                                                  \
			front = end_ ? INFINITY : slab_front(sn_, zlo, sw); front1 = end_ ? INFINITY : slab_front(sn_ + 1, zlo, sw);                          \
			s_front = sn_;                                                                                                                        \
		}
		const int n_batches = (n_list + 63) >> 6;
		int slab_next = 0, slab_cur = 0;
		uint32_t e_cur = entry(0, &slab_cur);
		uint32_t e_next = n_batches > 1 ? entry(1, &slab_next) : NONE;
		for (int b = 0; b < n_batches; ++b) {
			const bool more = b + 1 < n_batches;
			const int slab_after = more ? (sorted ? slab_next : s_front) : N_SLABS + 1;   // slab of the first face after this batch
			const uint32_t e_lane = e_cur;
			if (more) {
				e_cur = e_next;
				e_next = b + 2 < n_batches ? entry(b + 2, &slab_next) : NONE;   // (the entries two batches ahead are on their way while this one is evaluated)
			}
			// a pixel that holds its K nearest (and its colour) in front of everything from this batch on needs nothing more
			const bool fin = !in_img || ((!want_sil || c_lt >= K) && (!want_rgb || bz < front));
			const bool need = (early ? !fin : in_img) && !FIND_ABL(a.ablate, 4);
			const int nb = stage(e_lane);
			const int slab_v = (int)(e_lane >> 24);   // (sorted lists are binned: position in the batch = lane)
			// positions of the batch at which a new slab begins (bit p: entry p starts one), plus the position behind the batch's last entry
			const unsigned long long chg = sorted ? (__ballot(lane > 0 && lane < nb && slab_v != __shfl_up(slab_v, 1, 64)) | (nb < 64 ? 1ull << nb : 0ull)) : 0ull;
			for (int t = 0; 2 * t < nb; ++t) {   // two faces per turn, in packed arithmetic
				// the front moves with every PAIR behind which a new slab begins (round 5; per batch of 64 before): the faces behind this pair
				// start at the slab of entry 2 t + 2 -- a band is then one slab wide, not one batch deep.  (Scalar bit tests per pair; the
				// band bookkeeping runs only where a slab begins.)
				{
					int s_new = s_front;
					if (2 * t + 2 >= nb) s_new = slab_after;
					else if ((chg >> (2 * t + 1)) & 3ull) s_new = __builtin_amdgcn_readlane(slab_v, 2 * t + 2);
					if (s_new > s_front) RASTER_ADVANCE(s_new)
				}
				const float* blk = &rec[wave][t * PAIR_STRIDE];
				const bool two = 2 * t + 1 < nb;   // (an odd batch: the last block's second half is stale, and masked out)
				// nobody who still needs faces lies inside either bbox: next (the far side of a closed surface goes by like this)
				const float4 bx4 = *reinterpret_cast<const float4*>(blk + 56), by4 = *reinterpret_cast<const float4*>(blk + 60);
				const bool inb_a = need & (px <= bx4.z) & (px >= bx4.x) & (py <= by4.z) & (py >= by4.x);
				const bool inb_b = two & need & (px <= bx4.w) & (px >= bx4.y) & (py <= by4.w) & (py >= by4.y);
				if (__ballot(inb_a | inb_b) == 0ull) continue;
				if (FIND_ABL(a.ablate, 64)) n_eval += two ? 2 : 1;
				Frag2 fr2;
				eval_pair(blk, px, py, &fr2);   // (every lane: the ones outside the bboxes compute along and are masked out below)
				const float2 fid = *reinterpret_cast<const float2*>(blk + 54);   // the two face indices
#pragma unroll
				for (int u = 0; u < 2; ++u) {
					const bool inb = u ? inb_b : inb_a;
					const bool f_inside = u ? fr2.inside_b : fr2.inside_a;
					const float f_pzc = u ? fr2.pz_clip.y : fr2.pz_clip.x, f_pz = u ? fr2.pz.y : fr2.pz.x, f_dist = u ? fr2.dist.y : fr2.dist.x;
					if (want_sil) {
						const bool cand = inb & (f_pzc >= 0.f) & (f_inside | (f_dist < blur));
						if (cand) {
							const float q = 1.0f - silhouette_prob(f_inside ? -f_dist : f_dist, inv_sigma);
							const bool lt = f_pzc < front, la = !lt & (f_pzc < front1), lb = !(lt | la);
							c_lt += lt ? 1 : 0; c_a += la ? 1 : 0; c_b += lb ? 1 : 0;
							a_lt *= lt ? q : 1.0f; a_a *= la ? q : 1.0f; a_b *= lb ? q : 1.0f;
						}
					}
					if (want_rgb) {
						// nearest inside fragment; equal depths: the lower face index (PyTorch3D's insertion order)
						const int f_id = __float_as_int(u ? fid.y : fid.x);
						if (inb & f_inside & (f_pz >= 0.f) & ((f_pz < bz) | ((f_pz == bz) & (f_id < bf)))) {
							bz = f_pz; bf = f_id; bd = -f_dist;
							bw0 = u ? fr2.w0.y : fr2.w0.x; bw1 = u ? fr2.w1.y : fr2.w1.x; bw2 = u ? fr2.w2.y : fr2.w2.x;
						}
					}
				}
			}
			if (nb == 0 && slab_after > s_front) RASTER_ADVANCE(slab_after)   // (a batch of the no-room path in which no face reaches the tile)
			wave_lds_sync();   // the records are overwritten by the next batch
			RASTER_CHECKPOINT()
			if (early && more) {
				const bool fin2 = !in_img || ((!want_sil || c_lt >= K) && (!want_rgb || bz < front));
				if (__ballot(!fin2) == 0ull) break;
			}
		}

#undef RASTER_ADVANCE
#undef RASTER_CHECKPOINT
		float alpha = 1.0f, thr = INFINITY;
		int tie = -1;
		if (want_sil) {
			alpha = a_lt * a_a * a_b;   // fewer than K candidates: all of them
			if (crossed) {
				// exactly K in front of f_at: their product, and the bound keeps everything from f_at on out of the backward.  More than K:
				// provisional (what an unresolved pixel -- more band candidates than the scratch holds -- is left with)
				alpha = a_at;
				thr = __uint_as_float(__float_as_uint(f_at) - 1u);   // the largest float in front of f_at (FLT_MAX for +inf)
			}
			const bool hard = in_img && crossed && n_at > K && !FIND_ABL(a.ablate, 2);
			const unsigned long long hard_m = __ballot(hard);
			if (hard_m) {
				// ---- second sweep: the band candidates of the hard pixels.  Faces of slab s have their fragments in [edge of s, front of s + 2):
				// a band [front of s_prev, front of s_at) is reached by the slabs s_prev - 1 .. s_at - 1.  A hard pixel needs a tenth of the
				// tile's list (three slabs, inside its own blur margin), and the hard pixels of a tile need different tenths: evaluated tile-wide
				// -- every lane its pixel, a face at a time -- the sweep cost half of the first one.  So the work is COMPACTED: the (pixel, face)
				// pairs that pass the cheap tests (slab range, blurred bbox) are queued in LDS and evaluated 64 at a time, one pair per lane,
				// whichever pixel and face it is; a candidate inside its pixel's band goes to that pixel's column of the wave's scratch.
				int s_lo = hard ? s_prev - 1 : 0x7FFFFFFF, s_hi = hard ? s_at - 1 : -0x7FFFFFFF;
#pragma unroll
				for (int d = 1; d < 64; d <<= 1) { s_lo = min(s_lo, __shfl_xor(s_lo, d, 64)); s_hi = max(s_hi, __shfl_xor(s_hi, d, 64)); }
				s_lo = __builtin_amdgcn_readfirstlane(s_lo); s_hi = __builtin_amdgcn_readfirstlane(s_hi);
				int b_first = 0, b_last = n_batches - 1;
				if (sorted) {   // (a sorted list has at most BIN_CAP / 64 = 32 batches: one lane per batch looks at its first entry)
					const int s0 = lane < n_batches ? (int)(lp[lane * 64] >> 24) : 0x7FFFFFFF;
					b_first = (int)__popcll(__ballot(lane >= 1 && lane < n_batches && s0 < s_lo));   // batch b goes by when batch b + 1 still starts in front of s_lo
					b_last = (int)__popcll(__ballot(lane < n_batches && s0 <= s_hi)) - 1;
				}
				int n_eval2 = 0, n_staged2 = 0, n_items2 = 0;
				// this lane's own slab range (an unsorted list carries no slabs: every face)
				const int my_lo = sorted ? s_prev - 1 : -0x7FFFFFFF, my_hi = sorted ? s_at - 1 : 0x7FFFFFFF;
				float* const recw = &rec[wave][0];                       // records of this sweep: FLAT, piece k of record j at [REC_DW j + 4 (k ^ (j & 7))] (the swizzle spreads the 16-byte staging writes of neighbouring lanes over the banks)
				unsigned* const bcnt = &band_n[wave][0];                 // [l]: band candidates of lane l's pixel so far; [64 + l], [128 + l]: bits of their smallest / largest depth
				unsigned short* const queue = &item_q[wave][0];          // ring of 128 items
				bcnt[lane] = 0u; bcnt[64 + lane] = 0x7F800000u; bcnt[128 + lane] = 0u;
				int q_head = 0, q_n = 0;
				for (int b = b_first; b <= b_last; ++b) {
					int dummy;
					uint32_t e = entry(b, &dummy);
					if (sorted && e != NONE && ((int)(e >> 24) < s_lo || (int)(e >> 24) > s_hi)) e = NONE;   // only the faces of the slabs some hard pixel needs
					// stage flat: the record's 16-byte pieces, the list entry (slab << 24 | face) in its id slot
					const unsigned long long m_have = __ballot(e != NONE);
					const int nb = (int)__popcll(m_have);
					if (e != NONE) {
						const int pos = (int)__popcll(m_have & ((1ull << lane) - 1ull));
						const float4* src = reinterpret_cast<const float4*>(rp_img + (e & FACE_MASK));
						float4* d = reinterpret_cast<float4*>(recw + pos * REC_DW);
#pragma unroll
						for (int k = 0; k < REC_F4; ++k) { float4 q = src[k]; if (k == 6) q.w = __int_as_float((int)e); d[k ^ (pos & 7)] = q; }   // (the list entry in the id slot: slab << 24 | face)
					}
					wave_lds_sync();
					n_staged2 += nb;
					int j = 0;
					for (;;) {
						// queue (pixel, face) pairs until 64 wait or the batch's faces are through
						for (; j < nb && q_n < 64; ++j) {
							const float4 bb = *reinterpret_cast<const float4*>(recw + j * REC_DW + 4 * (7 ^ (j & 7)));   // xmin xmax ymin ymax
							const int sl = (int)((uint32_t)__float_as_int(recw[j * REC_DW + 4 * (6 ^ (j & 7)) + 3]) >> 24);
							const bool want = hard & (sl >= my_lo) & (sl <= my_hi) & (px <= bb.y) & (px >= bb.x) & (py <= bb.w) & (py >= bb.z);
							const unsigned long long wm = __ballot(want);
							if (wm == 0ull) continue;
							if (want) queue[(q_head + q_n + (int)__popcll(wm & ((1ull << lane) - 1ull))) & 127] = (unsigned short)((lane << 8) | j);
							q_n += (int)__popcll(wm);
						}
						if (q_n == 0) break;
						wave_lds_sync();
						// evaluate up to 64 of them, one per lane (the records are overwritten by the next batch: a batch drains its queue)
						const int count = min(q_n, 64);
						const bool act = lane < count;
						const unsigned it = (unsigned)queue[(q_head + lane) & 127];
						const int hl = act ? (int)(it >> 8) : lane, jj = act ? (int)(it & 255u) : 0;
						const int xh = tile_x * T8 + (hl & 7), yh = tile_y * T8 + (hl >> 3);
						const float pxh = 1.0f - (2.0f * xh + 1.0f) / (float)W, pyh = 1.0f - (2.0f * yh + 1.0f) / (float)H;   // (the owner's px, py to the bit: the same expressions)
						FaceRec r = load_rec(reinterpret_cast<const float4*>(recw + jj * REC_DW), jj & 7);
						r.f &= (int)FACE_MASK;
						Frag fr;
						eval_core(r, pxh, pyh, &fr);
						// the band of the item's pixel lives in that pixel's lane
						const float lo_h = __shfl(f_prev, hl, 64), hi_h = __shfl(f_at, hl, 64);
						const bool cand = act & (fr.pz_clip >= 0.f) & (fr.inside | (fr.dist < blur)) & (fr.pz_clip >= lo_h) & (fr.pz_clip < hi_h);
						if (cand) {
							// (lanes that deliver to the same pixel in one instruction are served in lane order: the slot order is the queue order)
							const unsigned slot = __hip_atomic_fetch_add(&bcnt[hl], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
							const unsigned zb = __float_as_uint(fr.pz_clip + 0.0f);
							__hip_atomic_fetch_min(&bcnt[64 + hl], zb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
							__hip_atomic_fetch_max(&bcnt[128 + hl], zb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
							if (slot < (unsigned)KN_CAP && !FIND_ABL(a.ablate, 1)) {
								const int64_t o = (int64_t)slot * 64 + hl;
								wz0[o] = fr.pz_clip;
								wq0[o] = 1.0f - silhouette_prob(fr.inside ? -fr.dist : fr.dist, inv_sigma);
								wf0[o] = r.f;
							}
						}
						n_items2 += count;
						q_head = (q_head + count) & 127; q_n -= count;
						++n_eval2;
						wave_lds_sync();
					}
				}
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the band candidates other lanes delivered to this lane's column have left the wave
				const int nbnd = (int)bcnt[lane];
				const float z_lo = __uint_as_float(bcnt[64 + lane]), z_hi = __uint_as_float(bcnt[128 + lane]);
				wave_lds_sync();
				const int rank = K - n_prev;   // 1-based rank of the K-th nearest inside the band
				const bool solve = hard && nbnd <= KN_CAP && nbnd > rank && !FIND_ABL(a.ablate, 1);   // (nbnd == n_at - n_prev > rank by construction)
				{
					const unsigned long long trunc = __ballot(hard && nbnd > KN_CAP);
					int wave_max = hard ? nbnd : 0;   // diagnostics: [5] largest band seen, [26] band candidates in all
					if (FIND_ABL(a.ablate, 64)) {
						int tb2 = hard ? nbnd : 0;
#pragma unroll
						for (int d = 1; d < 64; d <<= 1) tb2 += __shfl_xor(tb2, d, 64);
						if (lane == 0) atomicAdd(&a.flags[26], tb2);
					}
#pragma unroll
					for (int d = 1; d < 64; d <<= 1) wave_max = max(wave_max, __shfl_xor(wave_max, d, 64));
					if (lane == 0) {   // diagnostics: [4] pixels that needed the second sweep; [1] of those, left unresolved
						atomicAdd(&a.flags[4], (int)__popcll(hard_m));
						atomicMax(&a.flags[5], wave_max);
						if (trunc) atomicAdd(&a.flags[1], (int)__popcll(trunc));
						if (FIND_ABL(a.ablate, 64)) {   // [27] second-sweep evaluations (64 pairs each), [28] tiles that took it, [29] (pixel, face) pairs queued, [30] faces staged, [31] the tiles' list lengths
							atomicAdd(&a.flags[27], n_eval2); atomicAdd(&a.flags[28], 1); atomicAdd(&a.flags[29], n_items2); atomicAdd(&a.flags[30], n_staged2);
							atomicAdd(&a.flags[31], n_list);
						}
					}
				}
				if (solve) {
					// (the column was written by other lanes of this wave, the stores have been waited for; the loads go past the CU's L1 -- sc1 --
					// which may still hold this address from an earlier tile)
					const int n = nbnd;
					auto ldz = [&](int i) __attribute__((always_inline)) { return __hip_atomic_load(&wz[(int64_t)i * 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
					auto ldq = [&](int i) __attribute__((always_inline)) { return __hip_atomic_load(&wq[(int64_t)i * 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
					auto ldf = [&](int i) __attribute__((always_inline)) { return __hip_atomic_load(&wf[(int64_t)i * 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
					auto scan_z = [&](auto&& fn) {   // one pass over the lane's band depths, KU loads in flight, fn(bits of the depth)
						int i = 0;
						for (; i + KU <= n; i += KU) {
							float v[KU];
#pragma unroll
							for (int u = 0; u < KU; ++u) v[u] = ldz(i + u);
#pragma unroll
							for (int u = 0; u < KU; ++u) fn(__float_as_uint(v[u] + 0.0f));
						}
						for (; i < n; ++i) fn(__float_as_uint(ldz(i) + 0.0f));
					};
					unsigned lo = __float_as_uint(z_lo + 0.0f), hi = __float_as_uint(z_hi + 0.0f);
					// Radix search for the rank-th smallest depth (non-negative floats order like their bit patterns).  Invariant: every band
					// candidate lies in [lo, hi] or was counted in c_lo (in front of lo) or lies behind hi; the rank-th smallest is inside
					// [lo, hi].  A level histograms (z - lo) >> shift into 32 bins in ONE read of the band and keeps the bin that holds the
					// rank-th; once that bin has at most 4 candidates they are fetched and ranked directly.
					int c_lo = 0;
					// the lane's 32 counters (16 bits each) live in LDS: hist[w * 64 + lane], w = bin >> 1
					unsigned* const hist = reinterpret_cast<unsigned*>(&rec[wave][0]) + lane;
					while (lo < hi) {
						const unsigned span = hi - lo;
						const int shift = span < 32u ? 0 : (27 - __builtin_clz(span));  // (span >> shift) <= 31
#pragma unroll
						for (int w = 0; w < 16; ++w) hist[w * 64] = 0u;
						scan_z([&](unsigned zb) {
							if (zb < lo || zb > hi) return;
							const unsigned bin = (zb - lo) >> shift;
							__hip_atomic_fetch_add(hist + (bin >> 1) * 64, 1u << ((bin & 1u) * 16u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
						});
						const int need_r = rank - c_lo;
						int acc = 0, sel = 31, in_sel = 0;
						bool found = false;
#pragma unroll
						for (int w = 0; w < 16; ++w) {
							const unsigned hw = hist[w * 64];
#pragma unroll
							for (int h = 0; h < 2; ++h) {
								const int cb = (int)((hw >> (h * 16)) & 0xFFFFu);
								if (!found) {
									if (acc + cb >= need_r) { sel = 2 * w + h; in_sel = cb; found = true; }
									else acc += cb;
								}
							}
						}
						c_lo += acc;
						lo = lo + ((unsigned)sel << shift);
						hi = min(hi, lo + ((1u << shift) - 1u));
						if (shift == 0) break;  // a single depth value is left
						if (in_sel <= 4) {
							// fetch the (at most 4) candidates of the bin and take the (rank - c_lo)-th smallest of them
							unsigned c0 = 0xFFFFFFFFu, c1 = 0xFFFFFFFFu, c2 = 0xFFFFFFFFu, c3 = 0xFFFFFFFFu;
							int mm = 0;
							scan_z([&](unsigned zb) {
								// (selects, not an if-chain: the compiler turned that into a four-element array in scratch)
								const bool in = zb >= lo && zb <= hi;
								c0 = (in && mm == 0) ? zb : c0; c1 = (in && mm == 1) ? zb : c1;
								c2 = (in && mm == 2) ? zb : c2; c3 = (in && mm == 3) ? zb : c3;
								mm += in ? 1 : 0;
							});
							unsigned t;
							if (c0 > c1) { t = c0; c0 = c1; c1 = t; }
							if (c2 > c3) { t = c2; c2 = c3; c3 = t; }
							if (c0 > c2) { t = c0; c0 = c2; c2 = t; }
							if (c1 > c3) { t = c1; c1 = c3; c3 = t; }
							if (c1 > c2) { t = c1; c1 = c2; c2 = t; }
							const int rnk = rank - c_lo;  // 1-based rank inside the bin
							const unsigned T = rnk == 1 ? c0 : (rnk == 2 ? c1 : (rnk == 3 ? c2 : c3));
							c_lo += (c0 < T) + (c1 < T) + (c2 < T) + (c3 < T);
							lo = hi = T;
							break;
						}
					}
					// lo = bits of the K-th nearest depth, c_lo = band candidates strictly in front of it
					const int keep = rank - c_lo;   // how many of the candidates AT that depth belong to the K nearest (>= 1)
					int n_tie = 0;
					float a_sel = 1.0f, a_tie = 1.0f;
					unsigned nxt = 0x7F800000u;   // bits of the nearest depth BEHIND the K-th
					{
						int i = 0;
						for (; i + KU / 2 <= n; i += KU / 2) {
							float zv[KU / 2], qv[KU / 2];
#pragma unroll
							for (int u = 0; u < KU / 2; ++u) { zv[u] = ldz(i + u); qv[u] = ldq(i + u); }
#pragma unroll
							for (int u = 0; u < KU / 2; ++u) {
								const unsigned zb = __float_as_uint(zv[u] + 0.0f);
								a_sel *= zb < lo ? qv[u] : 1.0f; a_tie *= zb == lo ? qv[u] : 1.0f; n_tie += zb == lo ? 1 : 0;
								nxt = zb > lo ? min(nxt, zb) : nxt;
							}
						}
						for (; i < n; ++i) {
							const unsigned zb = __float_as_uint(ldz(i) + 0.0f);
							const float qv = ldq(i);
							a_sel *= zb < lo ? qv : 1.0f; a_tie *= zb == lo ? qv : 1.0f; n_tie += zb == lo ? 1 : 0;
							nxt = zb > lo ? min(nxt, zb) : nxt;
						}
					}
					const float zk = __uint_as_float(lo);
					if (n_tie <= keep) {
						alpha = a_prev * a_sel * a_tie;
						// the bound the backward compares a candidate's depth with: the MIDPOINT between the K-th depth and the next one behind it
						// (in the band, else the band's far edge)
						thr = fminf(0.5f * (zk + __uint_as_float(nxt)), thr);
					} else {
						// more candidates AT the K-th depth than fit: PyTorch3D's K-buffer keeps the lower face indices (insertion order; a later
						// fragment of EQUAL depth does not displace an earlier one).  Ties are not exotic: a pixel outside a fan of faces that share
						// their nearest vertex gets that vertex's depth from every one of them.  The `keep` lowest ids, multiplied in id order.
						int last = -1;
						float a_k = 1.0f;
						for (int k = 0; k < keep; ++k) {
							int best = 0x7FFFFFFF;
							float bq = 1.0f;
							for (int i = 0; i < n; ++i) {
								if (__float_as_uint(ldz(i) + 0.0f) != lo) continue;
								const int id = ldf(i);
								if (id > last && id < best) { best = id; bq = ldq(i); }
							}
							last = best; a_k *= bq;
						}
						alpha = a_prev * a_sel * a_k;
						thr = -zk;     // negative: "candidates tied at this depth are decided by tie_face" (the last face kept)
						tie = last;
					}
				}
			}
		}

		if (FIND_ABL(a.ablate, 64)) {  // diagnostics: [24] (pixel, face) tests issued (64-lane slots, in units of 64), [25] silhouette candidates seen (units of 64)
			int te = n_eval, tc = in_img ? c_lt + c_a + c_b : 0;
#pragma unroll
			for (int d = 1; d < 64; d <<= 1) { te += __shfl_xor(te, d, 64); tc += __shfl_xor(tc, d, 64); }
			if (lane == 0 && te) { atomicAdd(&a.flags[24], (te + 32) >> 6); atomicAdd(&a.flags[25], (tc + 32) >> 6); }
		}
		if (in_img) {
			const int64_t pix = ((int64_t)img * H + yi) * W + xi;
			if (want_sil) {
				a.mask[pix] = 1.0f - alpha;
				a.zthr[pix] = thr;
				a.alpha_ws[pix] = alpha;
				if (thr < 0.f) a.tie_face[pix] = tie;
			}
			if (want_rgb) {   // the nearest inside fragment: shade_kernel turns it into the pixel's colour
				a.p2f_ws[pix] = bf;
				*reinterpret_cast<float4*>(a.frag_ws + pix * 8) = make_float4(bw0, bw1, bw2, bz);
				a.frag_ws[pix * 8 + 4] = bd;
			}
		}
		wave_lds_sync();  // the next tile reuses the record area
	}
}
