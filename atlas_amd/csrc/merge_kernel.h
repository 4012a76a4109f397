// merge_kernel.h -- merge + exact rescoring of one query's candidates, as a device function: it is the body of
// merge_rescore_kernel (atlas_hip.hip) and the tail of the coop scan's last workgroups (scan_kernel.h). Included by scan_kernel.h behind
// the helpers it uses (query_eps, prune_threshold, ...); NT threads = one workgroup, `smem` = its dynamic LDS (>= ScanPlan::merge_lds).
#pragma once

#define MERGE_SMAX 2048          // max candidates rescored per query in the merge
#define MERGE_GMAX 1024          // most scan workgroups (= threads of a merge block)
#define MERGE_HEAD 8             // entries of every list requested speculatively together with its length

// Wave shuffles whose lane index is formed AT THE USE (volatile v_mbcnt): inside the scan kernel the merge sits in a loop, and the index
// registers of __shfl_* -- (lane ^ o) << 2 for every distance, pure functions of the lane id -- would be hoisted out of it and, at the
// 128-register cap, spilled (a private segment costs every launch ~12 us). Same ds_bpermute, four more VALU instructions per shuffle.
static __device__ __forceinline__ int lane_now() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
static __device__ __forceinline__ uint32_t bperm_u32(const int src_lane, const uint32_t v) {
    return (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)v);      // the hardware takes the lane index modulo 64
}
static __device__ __forceinline__ uint32_t shfl_xor_n(const uint32_t v, const int o) { return bperm_u32(lane_now() ^ o, v); }
static __device__ __forceinline__ float shfl_xor_n(const float v, const int o) { return bits_f32(bperm_u32(lane_now() ^ o, f32_bits(v))); }
static __device__ __forceinline__ double shfl_xor_n(const double v, const int o) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const int src = lane_now() ^ o;
    const unsigned long long r = (unsigned long long)bperm_u32(src, (uint32_t)b) | ((unsigned long long)bperm_u32(src, (uint32_t)(b >> 32)) << 32);
    return __builtin_bit_cast(double, r);
}
static __device__ __forceinline__ uint32_t shfl_up_n(const uint32_t v, const int o) { return bperm_u32(lane_now() - o, v); }     // lanes < o: garbage (callers ignore it)
static __device__ __forceinline__ uint32_t shfl_down_n(const uint32_t v, const int o) { return bperm_u32(lane_now() + o, v); }   // lanes >= 64 - o: garbage (ditto)

// ------------------------------------------------------------------------------------------
// merge + exact rescoring: one block per query
// ------------------------------------------------------------------------------------------
struct MergeParams {
    const uint16_t* slab; int64_t N; int d;
    const void* q; int q_dtype, qbase;  // the queries [.][d] (the caller's, or the fp16 rows of the sample pass): block b converts row qbase + b itself (RNE to fp16 = `.half()`)
    float pmax;                  // eps = GAMMA |q| pmax, as in the scan
    int pmax_trusted;            // the scan took pmax as certified and measured no norms (ATLAS_SCAN_TRUST_PMAX): every row rescored here is held
                                 // to it -- a larger one raises ATLAS_F_PMAX_VIOLATION (the caller's certificate was stale)
    const uint2* lists; const uint32_t* list_cnt; const uint32_t* wg_stat; int G, cap;   // the scan's per-(workgroup, query) candidate lists
    int total_cap;               // most candidates a query may bring to the merge (more: exact path)
    uint32_t* epoch;             // per-workspace call counter: block 0 bumps it (the next scan's granule tag)
    uint32_t* ticket;            // the scan's pool-tile counter: block 0 puts it back to zero for the next scan
    uint32_t* qflag;             // read, then cleared for the next call by the block that owns the query
    int k, q0;                   // q0: first query of this chunk (output row offset)
    int key_cap;                 // approximate-score keys that fit in LDS
    unsigned long long* dbg;     // optional per-phase cycle stamps of block 0 (tuning only; null in production)
    uint16_t* out_score; int64_t* out_idx; int32_t* out_status;
    uint64_t* out_packed; int64_t id_mul, id_add;   // optional: the winners also leave as cross-shard packed candidates (atlas_scan_topk_pack)
    // paired passes (ScanParams: grid.y == 2): blocks with blockIdx.y == 1 merge the nq2 queries of the second chunk, whose scan state /
    // lists live pair_state / pair_bulk bytes behind the first chunk's
    int nq1, nq2;
    size_t pair_state, pair_bulk;
    // FLAT mode (gscan_kernel.h): the candidates of query q are ONE contiguous list, lists[q * G * cap ...], of flat_cnt[q] entries (clipped
    // to G * cap); it is cut into G virtual segments of cap entries so that the gather below works on it unchanged. nstat = entries of
    // wg_stat (the scan's workgroups); 0 = G.
    const uint32_t* flat_cnt;
    int nstat;
    int plan_word;               // ATLAS_ST_PLAN of the call (the same in every pass), written by block 0
};

// Canonical exact score (common.h exact_dot_f16) computed by ONE WAVE: lane j is chain j and adds the
// products of elements j, j+64, ... in order; the chains are combined by an xor butterfly (distance 1,
// 2, .. 32), which is the canonical balanced tree because IEEE addition is commutative. Every lane
// returns the same double. Loads are 2 B per lane, 128 B contiguous per wave access.
//   qs: query (fp16 bits) in LDS or global; prow: slab row in global memory
static __device__ __forceinline__ double wave_exact_dot(const uint16_t* __restrict__ qs,
                                                        const uint16_t* __restrict__ prow, const int d, const int lane) {
    double c = 0.0;
    if (d == D_FAST) {
        uint16_t pv[D_FAST / 64];
#pragma unroll
        for (int i = 0; i < D_FAST / 64; ++i) pv[i] = prow[i * 64 + lane];          // all 12 loads in flight
#pragma unroll
        for (int i = 0; i < D_FAST / 64; ++i)
            c += (double)(float)__builtin_bit_cast(_Float16, qs[i * 64 + lane]) *
                 (double)(float)__builtin_bit_cast(_Float16, pv[i]);
    } else {
        for (int e = lane; e < d; e += 64)
            c += (double)(float)__builtin_bit_cast(_Float16, qs[e]) * (double)(float)__builtin_bit_cast(_Float16, prow[e]);
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) c += shfl_xor_n(c, m);
    return c;
}

// one block per query. Steps: (1) gather the approximate-score keys of every workgroup's list into
// LDS; (2) k-th largest by a greedy bit search that starts at the first bit where the keys differ and
// stops 2^-15 (relative) short of exact -- any lower bound of the k-th is a valid T; (3) candidate band
// s~ > prune_threshold(T); (4) exact rescoring in the canonical order: rows staged in LDS with coalesced
// loads, 8 lanes per candidate (lane j = chain j), chains combined by the canonical tree; (5) rank.
template <int NT>
static __device__ __forceinline__ void merge_rescore_body(const MergeParams& p, const int q, unsigned char* __restrict__ smem) {
    // layout: qs[d] u16 (padded to 16 B) | misc[64] | s_row[SMAX] | s_app[SMAX] | s_key[SMAX] u64 | s_off[MERGE_GMAX + 8] | keys[key_cap]
    uint16_t* qs = (uint16_t*)smem;
    const int qbytes = ((p.d * 2 + 15) / 16) * 16;
    uint32_t* misc = (uint32_t*)(smem + qbytes);   // [1] kmax [2] kmin [4] maxerr bits [5] nsurv [6] scan flags [7] pmax^2 bits [9] largest offending row norm^2 (trusted pmax) [16..31] wave totals of the offset scan
    uint32_t* s_row = misc + 64;
    float* s_app = (float*)(s_row + MERGE_SMAX);
    uint64_t* s_key = (uint64_t*)(s_app + MERGE_SMAX);
    uint32_t* s_off = (uint32_t*)(s_key + MERGE_SMAX);            // [G + 1] exclusive offsets of the workgroups' segments
    uint32_t* keys = s_off + MERGE_GMAX + 8;

    // (the thread index enters through an opaque copy: inside the scan kernel this body sits in a loop -- the last workgroup may come
    //  round for abandoned queries -- and hipcc would otherwise hoist everything lane-derived out of it and spill it)
    int tid_o = (int)threadIdx.x;
    asm volatile("" : "+v"(tid_o));
    const int tid = tid_o, lane = tid & 63, wave = tid >> 6;
    const int k = p.k;
    uint16_t* o_score = p.out_score + (size_t)(p.q0 + q) * k;
    int64_t* o_idx = p.out_idx + (size_t)(p.q0 + q) * k;
    int32_t* o_qst = p.out_status + ATLAS_STATUS_HEADER + p.q0 + q;

    auto fallback = [&]() {
        if (tid == 0) {
            *o_qst = ATLAS_Q_FALLBACK;
            atomicOr((uint32_t*)&p.out_status[ATLAS_ST_FLAGS], (uint32_t)ATLAS_F_FALLBACK);
            atomicAdd((uint32_t*)&p.out_status[ATLAS_ST_N_FALLBACK], 1u);
        }
    };
    if (p.dbg && q == 0 && tid == 0) p.dbg[0] = wall_clock64();
    // (0) the segment table's inputs are requested FIRST, together with the query (one trip to memory instead of two: the barrier
    // behind the query conversion used to stand between them): thread g takes workgroup g's list length for this query and its
    // norm / flag word (the scan's certification state is reduced here, the scan kernel itself ends without a single global atomic)
    static_assert(NT >= MERGE_GMAX, "one thread per scan workgroup");
    uint32_t cg = 0, pmb = 0, flg = 0;
    const int nstat = p.nstat > 0 ? p.nstat : p.G;
    if (tid < p.G) {
        if (p.flat_cnt != nullptr) {
            const uint32_t n = p.flat_cnt[q], lo = (uint32_t)tid * (uint32_t)p.cap;
            cg = n > lo ? (n - lo < (uint32_t)p.cap ? n - lo : (uint32_t)p.cap) : 0u;
        } else {
            cg = p.list_cnt[(size_t)q * p.G + tid];
        }
    }
    if (tid < nstat) {
        pmb = p.wg_stat[(size_t)tid * 2 + 0];                         // non-negative floats order like their bits
        flg = p.wg_stat[(size_t)tid * 2 + 1];
    }
    // ... and, speculatively, the first MERGE_HEAD entries of that list (64 bytes; the length is not known yet): most lists of a small
    // shard are shorter than that (1M rows: 2.7 entries on average), and for those the gather below needs no trip of its own
    uint2 head[MERGE_HEAD];
    if (tid < p.G) {
        const uint4* hp = (const uint4*)(p.lists + ((size_t)q * p.G + tid) * p.cap);
#pragma unroll
        for (int u = 0; u < MERGE_HEAD / 2; ++u) {
            const uint4 v = hp[u];
            head[2 * u] = make_uint2(v.x, v.y);
            head[2 * u + 1] = make_uint2(v.z, v.w);
        }
    }
    // this block's query, converted here (no preparation kernel), and its certified error bound
    double ss = 0.0;
    for (int i = tid; i < p.d; i += NT) {
        const size_t off = (size_t)(p.qbase + q) * p.d + i;
        uint16_t h;
        if (p.q_dtype == ATLAS_DT_F16) h = ((const uint16_t*)p.q)[off];
        else if (p.q_dtype == ATLAS_DT_F32) h = __builtin_bit_cast(uint16_t, (_Float16)((const float*)p.q)[off]);
        else h = __builtin_bit_cast(uint16_t, (_Float16)bits_f32((uint32_t)((const uint16_t*)p.q)[off] << 16));
        qs[i] = h;
        const double v = (double)(float)__builtin_bit_cast(_Float16, h);
        ss += v * v;
    }
    if (tid < 64) misc[tid] = (tid == 2) ? 0xffffffffu : 0u;
    uint32_t flagged = 0u;
    if (tid == 0) {                      // per-call state kept in the workspace is put back for the next call here: this block is the only
        flagged = p.qflag[q];            // reader of its query's flag, and every scan workgroup has finished
        p.qflag[q] = 0u;
        if (q == 0) { *p.epoch = *p.epoch + 1u; *p.ticket = 0u; p.out_status[ATLAS_ST_PLAN] = p.plan_word; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += shfl_xor_n(ss, o);
    double* s_ss = (double*)(misc + 32);                             // [16] per-wave partial sums
    __syncthreads();
    if (lane == 0) s_ss[wave] = ss;
    if (tid == 0) misc[8] = flagged;
    // the segment table: a block-wide exclusive scan of the list lengths
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t a = shfl_xor_n(pmb, o);
        pmb = a > pmb ? a : pmb;
        flg |= shfl_xor_n(flg, o);
    }
    if (lane == 0 && wave * 64 < nstat) { atomicMax(&misc[7], pmb); if (flg) atomicOr(&misc[6], flg); }
    uint32_t inc = cg;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = shfl_up_n(inc, o);
        if (lane >= o) inc += y;
    }
    if (lane == 63) misc[16 + wave] = inc;
    __syncthreads();
    uint32_t wbase = 0;
    for (int w = 0; w < wave; ++w) wbase += misc[16 + w];
    if (tid < p.G) s_off[tid] = wbase + inc - cg;
    if (tid == p.G - 1) s_off[p.G] = wbase + inc;
    __syncthreads();
    const uint32_t total = s_off[p.G];
    // the scan flagged this query (band overflow), or it brings more candidates than the merge is sized for: exact path
    if (misc[8] != 0u || total > (uint32_t)p.total_cap) { fallback(); return; }
    // flat candidate index -> entry: binary search in the segment table (only band members and keys beyond the LDS cache need it)
    auto entry_at = [&](const uint32_t i) -> uint2 {
        int lo = 0, hi = p.G;                                         // s_off[lo] <= i < s_off[hi]
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_off[mid] <= i) lo = mid; else hi = mid; }
        return p.lists[((size_t)q * p.G + lo) * p.cap + (i - s_off[lo])];
    };
    // the first key_cap / 2 candidates are cached in LDS, key and row; a longer candidate list (poor initial threshold, adversarial
    // data, 32M-row shards) is re-read from the lists (L2) in the two later passes instead of being handed to the exact path
    const uint32_t kc2 = (uint32_t)p.key_cap / 2u;
    uint32_t* rows_l = keys + kc2;
    const uint32_t ncache = total < kc2 ? total : kc2;
    auto key_at = [&](const uint32_t i) -> uint32_t { return i < ncache ? keys[i] : f32_order_key(bits_f32(entry_at(i).x)); };
    if (p.dbg && q == 0 && tid == 0) p.dbg[1] = wall_clock64();
    // (1) keys (+ rows) -> LDS with min / max: a wave takes 16 segments at a time and requests all of them before it uses the first
    // (the segments were written by workgroups of every XCD: each is its own trip to memory). The first MERGE_HEAD entries of every
    // list came with the table (one 64-byte read per list by the list's own thread: 16 KiB per block; a wave-wide speculative read
    // of 64 entries per list, 128 KiB per block, was measured slower); this loop fetches what lies beyond them
    uint32_t kmax = 0, kmin = 0xffffffffu;
    constexpr int NWV = NT / 64, SEG = 16;
    if (tid < p.G) {                                  // the heads that came with the table
        const uint32_t base = s_off[tid];
        const uint32_t nh = cg < (uint32_t)MERGE_HEAD ? cg : (uint32_t)MERGE_HEAD;
#pragma unroll
        for (int j = 0; j < MERGE_HEAD; ++j)
            if ((uint32_t)j < nh) {
                const uint32_t key = f32_order_key(bits_f32(head[j].x));
                if (base + j < ncache) { keys[base + j] = key; rows_l[base + j] = head[j].y; }
                kmax = key > kmax ? key : kmax;
                kmin = key < kmin ? key : kmin;
            }
    }
    for (int g0 = wave * SEG; g0 < p.G; g0 += NWV * SEG) {
        uint32_t longest = 0;
#pragma unroll
        for (int u = 0; u < SEG; ++u)
            if (g0 + u < p.G) { const uint32_t n = s_off[g0 + u + 1] - s_off[g0 + u]; longest = n > longest ? n : longest; }
        for (uint32_t r0 = MERGE_HEAD; r0 < longest; r0 += 64) {       // entries beyond the heads
            uint2 sc[SEG];
#pragma unroll
            for (int u = 0; u < SEG; ++u) {
                const int g = g0 + u < p.G ? g0 + u : p.G - 1;          // clamped: loads stay unconditional
                const uint32_t n = s_off[g + 1] - s_off[g], j = r0 + lane;
                sc[u] = p.lists[((size_t)q * p.G + g) * p.cap + (j < n ? j : 0)];
            }
#pragma unroll
            for (int u = 0; u < SEG; ++u) {
                if (g0 + u >= p.G) continue;
                const uint32_t base = s_off[g0 + u], n = s_off[g0 + u + 1] - base, j = r0 + lane;
                if (j < n) {
                    const uint32_t key = f32_order_key(bits_f32(sc[u].x));
                    if (base + j < ncache) { keys[base + j] = key; rows_l[base + j] = sc[u].y; }
                    kmax = key > kmax ? key : kmax;
                    kmin = key < kmin ? key : kmin;
                }
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t a = shfl_xor_n(kmax, o), b = shfl_xor_n(kmin, o);
        kmax = a > kmax ? a : kmax;
        kmin = b < kmin ? b : kmin;
    }
    if (lane == 0) { atomicMax(&misc[1], kmax); atomicMin(&misc[2], kmin); }
    __syncthreads();
    if (p.dbg && q == 0 && tid == 0) p.dbg[2] = wall_clock64();
    float theta = neg_inf();
    double qss = 0.0;
    for (int w = 0; w < NT / 64; ++w) qss += s_ss[w];
    const float eps = query_eps((float)qss * 1.000001f, p.pmax);
    if (total >= (uint32_t)k) {
        // (2) a lower bound T of the k-th largest key from ONE 1024-bin histogram over [kmin, kmax]
        // (any T with count(keys >= T) >= k is valid; the bin width, (kmax-kmin)/1024, only widens the
        // candidate band by the few entries that share the k-th's bin)
        kmax = misc[1]; kmin = misc[2];
        const uint32_t span = kmax - kmin;
        const int shift = span >= 1024u ? (32 - __builtin_clz(span)) - 10 : 0;     // (key-kmin)>>shift < 1024
        uint32_t* hist = (uint32_t*)s_key;                                          // 4 KB of the (still unused) key area
        for (int i = tid; i < 1024; i += NT) hist[i] = 0;
        __syncthreads();
        for (uint32_t i = tid; i < total; i += NT) atomicAdd(&hist[(key_at(i) - kmin) >> shift], 1u);
        __syncthreads();
        if (tid < 64) {
            // lane l owns bins [16l, 16l+16); suffix sums across lanes find the lane, then the bin
            uint32_t own = 0;
#pragma unroll
            for (int b = 0; b < 16; ++b) own += hist[tid * 16 + b];
            uint32_t suf = own;                                   // inclusive suffix sum over lanes >= l
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t y = shfl_down_n(suf, o);
                if (tid + o < 64) suf += y;
            }
            const uint32_t above = suf - own;                     // keys in lanes > l
            if (above < (uint32_t)k && suf >= (uint32_t)k) {      // exactly one lane
                uint32_t acc = above; int bin = tid * 16;
                for (int b = 15; b >= 0; --b) {
                    acc += hist[tid * 16 + b];
                    if (acc >= (uint32_t)k) { bin = tid * 16 + b; break; }
                }
                misc[3] = kmin + ((uint32_t)bin << shift);        // lower edge of the bin holding rank k
            }
        }
        __syncthreads();
        const uint32_t prefix = misc[3];
        theta = prune_threshold(f32_from_order_key(prefix), eps);
    }
    if (p.dbg && q == 0 && tid == 0) p.dbg[3] = wall_clock64();
    // (3) candidate band (keys in LDS decide; only band members are re-read)
    const uint32_t theta_key = f32_order_key(theta);
    for (uint32_t i = tid; i < total; i += NT) {
        if (key_at(i) > theta_key) {
            const uint2 e = i < ncache ? make_uint2(f32_bits(f32_from_order_key(keys[i])), rows_l[i]) : entry_at(i);
            const uint32_t sidx = atomicAdd(&misc[5], 1u);
            if (sidx < MERGE_SMAX) { s_row[sidx] = e.y; s_app[sidx] = bits_f32(e.x); }
        }
    }
    __syncthreads();                                  // band complete; keys[] is dead from here on
    const uint32_t nsurv = misc[5];
    if (nsurv > MERGE_SMAX) { fallback(); return; }

    if (p.dbg && q == 0 && tid == 0) p.dbg[4] = wall_clock64();
    // (4) exact rescoring in the canonical order: one wave per candidate, 4 candidates per wave in flight
    // (their 48 two-byte row loads go out together: one HBM latency instead of four)
    if (p.d == D_FAST) {
        constexpr int NWV = NT / 64, R = 4;
        for (uint32_t i0 = wave; i0 < nsurv; i0 += NWV * R) {
            uint16_t pv[R][D_FAST / 64];
            uint32_t row[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint32_t i = i0 + r * NWV;
                row[r] = s_row[i < nsurv ? i : i0];                    // clamped: loads stay unconditional
                const uint16_t* prow = p.slab + (size_t)row[r] * D_FAST;
#pragma unroll
                for (int t = 0; t < D_FAST / 64; ++t) pv[r][t] = prow[t * 64 + lane];
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint32_t i = i0 + r * NWV;
                double c = 0.0;
                float n2 = 0.f;                                            // the row's squared norm (trusted-pmax check)
#pragma unroll
                for (int t = 0; t < D_FAST / 64; ++t) {
                    const float pf = (float)__builtin_bit_cast(_Float16, pv[r][t]);
                    c += (double)(float)__builtin_bit_cast(_Float16, qs[t * 64 + lane]) * (double)pf;
                    n2 = __builtin_fmaf(pf, pf, n2);
                }
#pragma unroll
                for (int m = 1; m < 64; m <<= 1) c += shfl_xor_n(c, m);   // canonical tree (common.h)
                if (p.pmax_trusted) {
#pragma unroll
                    for (int m = 1; m < 64; m <<= 1) n2 += shfl_xor_n(n2, m);
                    if (lane == 0 && i < nsurv && n2 > p.pmax * p.pmax) atomicMax(&misc[9], f32_bits(n2));   // non-negative floats order like their bits
                }
                if (lane == 0 && i < nsurv) {
                    s_key[i] = local_key(f64_to_f16_bits_rto(c), row[r]);
                    const float err = fabsf((float)((double)s_app[i] - c));   // a-posteriori check of the error model
                    const float ratio = eps > 0.f ? err / eps : (err > 0.f ? 2.0f : 0.0f);
                    atomicMax(&misc[4], f32_bits(ratio));
                }
            }
        }
    } else {
        for (uint32_t i = wave; i < nsurv; i += NT / 64) {
            const uint32_t row = s_row[i];
            const double sc = wave_exact_dot(qs, p.slab + (size_t)row * p.d, p.d, lane);
            if (lane == 0) {
                s_key[i] = local_key(f64_to_f16_bits_rto(sc), row);
                const float err = fabsf((float)((double)s_app[i] - sc));
                const float ratio = eps > 0.f ? err / eps : (err > 0.f ? 2.0f : 0.0f);
                atomicMax(&misc[4], f32_bits(ratio));
            }
        }
    }
    __syncthreads();
    if (p.dbg && q == 0 && tid == 0) p.dbg[5] = wall_clock64();
    // (5) rank by counting (keys are unique: the row is part of the key)
    for (uint32_t i = tid; i < nsurv; i += NT) {
        const uint64_t ki = s_key[i];
        uint32_t pos = 0;
        for (uint32_t j = 0; j < nsurv; ++j) pos += (s_key[j] > ki) ? 1u : 0u;
        if (pos < (uint32_t)k) {
            const uint16_t sc = f16_from_order_key((uint16_t)(ki >> 32));
            const uint32_t row = 0xffffffffu - (uint32_t)ki;
            o_score[pos] = sc;
            o_idx[pos] = (int64_t)row;
            if (p.out_packed) p.out_packed[(size_t)(p.q0 + q) * k + pos] = pack_candidate(sc, (uint64_t)((int64_t)row * p.id_mul + p.id_add));
        }
    }
    for (uint32_t i = nsurv + tid; i < (uint32_t)k; i += NT) {
        o_score[i] = 0xfc00; o_idx[i] = -1;
        if (p.out_packed) p.out_packed[(size_t)(p.q0 + q) * k + i] = 0ull;
    }
    if (p.dbg && q == 0 && tid == 0) p.dbg[6] = wall_clock64();
    if (tid == 0) {
        *o_qst = ATLAS_Q_OK;
        atomicAdd((uint32_t*)&p.out_status[ATLAS_ST_N_CANDIDATES], total);
        atomicAdd((uint32_t*)&p.out_status[ATLAS_ST_N_RESCORED], nsurv);
        atomicMax((uint32_t*)&p.out_status[ATLAS_ST_MAXERR_BITS], misc[4]);
        if (bits_f32(misc[4]) > 1.0f)
            atomicOr((uint32_t*)&p.out_status[ATLAS_ST_FLAGS], (uint32_t)ATLAS_F_EPS_VIOLATION);
        // scan-level flags / pmax (idempotent across blocks)
        if (misc[6]) atomicOr((uint32_t*)&p.out_status[ATLAS_ST_FLAGS], misc[6]);
        atomicMax((uint32_t*)&p.out_status[ATLAS_ST_PMAX_BITS], f32_bits(sqrtf(bits_f32(misc[7])) * 1.000001f));
        if (misc[9]) {           // a rescored row is longer than the bound the caller certified: the pruning margin of this call was too small
            atomicOr((uint32_t*)&p.out_status[ATLAS_ST_FLAGS], (uint32_t)ATLAS_F_PMAX_VIOLATION);
            atomicMax((uint32_t*)&p.out_status[ATLAS_ST_PMAX_BITS], f32_bits(sqrtf(bits_f32(misc[9]) * 1.001f)));
        }
    }
}

