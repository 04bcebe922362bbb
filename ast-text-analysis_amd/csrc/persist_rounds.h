// persist_rounds.h -- ALL remaining refinement rounds of a small domain in one launch.
//
// A round of window_sort.h is ~15 launches and two read-backs however few suffixes are left: the reference's own
// benchmark collection (analysis/utils.py:5-9, 100 identical strings) at n = 1000 -- 100 K symbols -- spent 4.4 ms in
// 124 launches, slower than one CPU core.  When the open domain has no more tiles than the device has CUs (one tile of
// LG_CHUNK positions per workgroup: 512 K elements; beyond that the large form at the end of this file) and no tie group
// is longer than a tile takes (LG_MAX_GROUP), every workgroup keeps its tile -- elements, group bounds, the slots they
// will fill -- in LDS and registers and runs prefix doubling (Manber-Myers / Larsson-Sadakane, as the doubling rounds of
// window_sort.h) to the end:  a round = gather the NAME of the suffix `depth` symbols further on (a name = the slot where
// a suffix's tie group starts, its own slot once it is placed), sort the tile in LDS by (group, name) with the wave64
// multisplit passes of lds_group_sort.h, read the new group bounds off the sorted keys, publish the new names, ONE grid
// barrier.  Names are double-buffered (a round reads one copy and writes the other: no workgroup can see a name of the
// round under way), so a round costs its sort plus one barrier instead of fifteen kernel boundaries and two host
// synchronisations.  The depth doubles per round: log2(longest repeat / first depth) rounds.
//
// The grid barrier: one monotonic counter -- every storing wave drains its stores, workgroup barrier, lane 0 releases
// (agent scope), adds and polls with relaxed agent-scope loads, s_sleep between polls, acquires, workgroup barrier.  What
// is handed from workgroup to workgroup is stored write-through and read past the L1 (pr_store / pr_load below: few
// dirty lines for the release to write back).  Every spin is bounded: a barrier that does not complete raises the abort
// flag and the host reports an internal error instead of hanging the device.  The host launches at most the number of
// workgroups the occupancy query admits at once and holds a lock from the launch to the read-back, so that two such
// kernels of one process never share the chip half resident each.
//
// LCP entries: the kernels write, per rank, what the two neighbours are KNOWN to share (the depth of the round in which
// they came apart); the host lists the slots of the domain and lvl0_lcp_text_list_kernel finishes the entries on the text
// once the suffix array stands (short ones directly, long ones through the irreducible-LCP finishing pass).
#pragma once
#include "lds_group_sort.h"

#define PR_MAX_ROUNDS 40                    // depth doubles per round: 2^32 symbols are covered long before
#define PR_CTL_BARRIER 0                    // ctl[]: the barrier's arrival counter,
#define PR_CTL_BAIL 1                       // a tie group longer than a tile takes (nothing was changed: the multi-launch rounds run),
#define PR_CTL_ROUNDS 2                     // rounds run,
#define PR_CTL_ABORT 3                      // a barrier timed out or the rounds did not end,
#define PR_CTL_OPEN 8                       // [PR_CTL_OPEN + r]: members of groups still open after round r
#ifdef PR_STAMPS                             // diagnostic build: block 0 stamps the phases of every round (wall clock, 10 ns)
#define PR_CTL_STAMPS (PR_CTL_OPEN + PR_MAX_ROUNDS)
#define PR_CTL_WORDS (PR_CTL_STAMPS + 2 * 8 * PR_MAX_ROUNDS)
#define PR_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) ((unsigned long long *)(a.ctl + PR_CTL_STAMPS))[round * 8 + (k)] = wall_clock64(); } while (0)
#else
#define PR_CTL_WORDS (PR_CTL_OPEN + PR_MAX_ROUNDS)
#define PR_STAMP(k) do { } while (0)
#endif
#ifndef PR_FENCE_MODE
#define PR_FENCE_MODE 3                     // 1: lane 0 acquires behind the barrier, 2: lane 0 releases in front of it, 3: both (0 / 1 / 2: diagnostic builds)
#endif
#define PR_SPIN_LIMIT (1u << 20)            // polls of a microsecond or two each before a barrier gives up (far beyond any round)

struct PrArgs {
    const u32 *elems, *gstart, *slots;      // the compacted domain (dc3_refine_compact_kernel): suffixes, group start flags, slots
    u32 m;
    u32 depth;                              // what the members of a group are known to share
    int name_bits;                          // bit_width(n - 1): a name is a slot
    u32 *name0, *name1;                     // names by text position, two copies (both initialised for every placed suffix)
    u32 *order_g;                           // the suffix array
    u32 *lcp_hint;                          // per slot of the domain: symbols the suffix is KNOWN to share with the one in front of it
                                            // (what its group shared when the seam opened; 0 at the first rank of a first group) -- or nullptr
    u32 *ctl;                               // PR_CTL_WORDS words, zeroed
};

// names of every suffix placed so far: its rank.  (alt != nullptr: the first domain of the separate placement pass --
// the suffix array holds the placed ranks only, the members of open groups sit in the sorted pairs)
__global__ __launch_bounds__(BLOCK) void persist_names_init_kernel(const u32 *__restrict__ order_g, const u32 *__restrict__ alt,
                                                                   const u64 *__restrict__ keep, u32 n,
                                                                   u32 *__restrict__ name0, u32 *__restrict__ name1)
{
    const u32 i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const u32 e = alt && ((keep[i >> 6] >> (i & 63u)) & 1u) ? alt[i] : order_g[i];
    name0[e] = i;
    name1[e] = i;
}

// What workgroups hand to each other inside the launch (the names; in the large form also elements and group bounds near
// a tile's ends) is stored write-through and read past the vector L1 (agent-scope relaxed atomics = `global_store /
// global_load ... sc1` on gfx950), AND lane 0 of every workgroup releases in front of the barrier and acquires behind
// it (MI355X_MICROARCH.md, "Valid forms").  Both were measured to be needed: with plain stores and loads every arrival's
// release wrote a 1 M-name round's worth of dirty lines back (a workgroup sharing its CU with one sitting in the barrier
// took 40 us over an 8 us round); with `sc1` / `sc0 sc1` accesses and NO fences the large form left the groups at tile
// boundaries unsorted -- a workgroup re-reads the group bounds around its tile's ends every round, and got the flags of
// two rounds ago although a workgroup on another XCD had stored to them since; one workgroup alone was always right,
// acquire alone was not enough, release + acquire by lane 0 is (a few dirty lines: the release is cheap now).
__device__ __forceinline__ void pr_store(u32 *p, u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u32 pr_load(const u32 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// All workgroups of the launch meet (barrier number `index`, 1-based).  False: the barrier was abandoned.
__device__ __forceinline__ bool pr_grid_barrier(u32 *ctl, u32 index, u32 *lds_flag)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every storing wave: its (write-through) stores are done
    __syncthreads();
    if (threadIdx.x == 0) {
#if PR_FENCE_MODE & 2
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        __hip_atomic_fetch_add(&ctl[PR_CTL_BARRIER], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const u32 target = gridDim.x * index;
        u32 spins = 0, ok = 1;
        while (__hip_atomic_load(&ctl[PR_CTL_BARRIER], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(4);
            if ((++spins & 63u) == 0u && __hip_atomic_load(&ctl[PR_CTL_ABORT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ok = 0; break; }
            if (spins > PR_SPIN_LIMIT) {
                __hip_atomic_store(&ctl[PR_CTL_ABORT], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
        }
#if PR_FENCE_MODE & 1
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        *lds_flag = ok;
    }
    __syncthreads();
    return *lds_flag != 0u;
}

// the last group start at or in front of tile position `local` (position 0 starts a group)
__device__ __forceinline__ u32 pr_group_start(const u64 *bits, u32 local)
{
    u32 wi = local >> 6;
    u64 x = bits[wi] & (((u64)2 << (local & 63u)) - 1ull);
    while (!x) x = bits[--wi];
    return wi * 64u + 63u - (u32)__builtin_clzll(x);
}

// MIN_WAVES = 4: one workgroup per CU, 116 registers a lane, nothing spilled -- what the host launches, for domains of at
// most as many tiles as the device has CUs (more tiles: the large form below).  (8: two per CU at 64 registers with a few
// spills; two workgroups share a CU's SIMDs without gaining on each other, and with the barrier's fences the form lost to
// the large one: no longer launched.)
template <int MIN_WAVES>
__global__ __launch_bounds__(LG_THREADS, MIN_WAVES) void refine_persist_kernel(PrArgs a)
{
    __shared__ LgLds lds;
    __shared__ u32 bar_flag, wg_count;
    const u32 lane = lane_id(), w = wave_id();
    const u32 base = blockIdx.x * LG_CHUNK;
    const u32 m = a.m;
    // ---- the tile: the groups that start in this workgroup's chunk (as refine_lds_sort_kernel takes them) ----
    {
        constexpr int PER = LG_WORDS / LG_WAVES;
        u32 gs[PER];
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const u64 p = (u64)base + (w + (u32)i * LG_WAVES) * 64u + lane;
            gs[i] = a.gstart[p < m ? p : (u64)m - 1u];
        }
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const u32 word = w + (u32)i * LG_WAVES;
            const u64 p = (u64)base + word * 64u + lane;
            const bool st = p < m ? gs[i] != 0u : p == m;
            const u64 bal = __ballot(st);
            if (lane == 0) lds.start_bits[word] = bal;
        }
    }
    __syncthreads();
    if (w == 0) {
        const u64 wv = lane < LG_CHUNK / 64 ? lds.start_bits[lane] : 0ull;
        const u64 nzb = __ballot(wv != 0ull);
        u32 begin_q = LG_NONE, end_q = LG_NONE;
        bool too_long = false;
        if (nzb) {
            const u32 fl = (u32)__ffsll((unsigned long long)nzb) - 1u, ll = 63u - (u32)__builtin_clzll(nzb);
            begin_q = fl * 64u + (u32)__builtin_ctzll(lds.start_bits[fl]);
            const u32 last_q = ll * 64u + 63u - (u32)__builtin_clzll(lds.start_bits[ll]);
            const u32 wi = ll + lane;
            u64 x = wi < LG_WORDS ? lds.start_bits[wi] : 0ull;
            if (lane == 0) x &= ~(((u64)2 << (last_q & 63u)) - 1ull);
            const u64 nb = __ballot(x != 0ull);
            u32 next_q = LG_NONE;
            if (nb) {
                const u32 l2 = (u32)__ffsll((unsigned long long)nb) - 1u;
                const u64 xw = ((u64)__shfl((u32)(x >> 32), l2, WAVE) << 32) | __shfl((u32)x, l2, WAVE);
                next_q = (ll + l2) * 64u + (u32)__builtin_ctzll(xw);
            }
            // (the multi-launch round leaves a longer last group to the global sort; here it ends the attempt)
            // (in the domain's last chunk the last "start" is position m itself: the tile ends there)
            const bool sentinel = (u64)base + last_q == (u64)m;
            too_long = !sentinel && (next_q == LG_NONE || next_q - last_q > LG_MAX_GROUP);
            end_q = sentinel || too_long ? last_q : next_q;
            if (begin_q == last_q && sentinel) begin_q = LG_NONE;       // (nothing but the sentinel)
        }
        if (lane == 0) {
            lds.hdr[0] = begin_q;
            lds.hdr[1] = begin_q == LG_NONE ? 0u : end_q - begin_q;
            if (too_long) __hip_atomic_store(&a.ctl[PR_CTL_BAIL], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    const u32 begin_q = lds.hdr[0], n_act = lds.hdr[1];
    const bool active = w * (LG_IPT * WAVE) < n_act;
    // ---- the tile's elements and slots (by tile position: a position keeps its slot, the elements move) ----
    u32 val[LG_IPT], slot_of[LG_IPT];
    u64 key[LG_IPT];
#pragma unroll
    for (int j = 0; j < LG_IPT; j++) {
        const u32 local = w * (LG_IPT * WAVE) + j * WAVE + lane;
        const bool in = local < n_act;
        const u64 r = (u64)base + begin_q + (in ? local : 0u);
        val[j] = in ? a.elems[r] : 0u;
        slot_of[j] = in ? a.slots[r] : 0u;
        const u32 st = in ? a.gstart[r] : 1u;           // (behind the tile: all starts)
        const u64 bal = __ballot(st != 0u);            // (the chunk-relative bits above were last read in front of the barrier)
        if (lane == 0) lds.start_bits[w * LG_IPT + j] = bal;
    }
    if (threadIdx.x == 0) lds.start_bits[LG_WORDS] = ~0ull;
    __syncthreads();
    // every member is named by the slot where its group starts (groups keep their stretches of the global order)
#pragma unroll
    for (int j = 0; j < LG_IPT; j++) {
        const u32 local = w * (LG_IPT * WAVE) + j * WAVE + lane;
        if (local >= n_act) continue;
        const u32 nm = slot_of[j] - (local - pr_group_start(lds.start_bits, local));
        pr_store(&a.name0[val[j]], nm);
        pr_store(&a.name1[val[j]], nm);
    }
    // the members of groups of two or more (the host hands over nothing else; counted all the same: the rounds end on it)
    u32 tile_open;
    {
        u32 open_cnt = 0;
#pragma unroll
        for (int j = 0; j < LG_IPT; j++) {
            const u32 local = w * (LG_IPT * WAVE) + j * WAVE + lane;
            if (local >= n_act) continue;
            const bool single = ((lds.start_bits[local >> 6] >> (local & 63u)) & 1ull) &&
                                ((lds.start_bits[(local + 1u) >> 6] >> ((local + 1u) & 63u)) & 1ull);
            open_cnt += single ? 0u : 1u;
        }
        if (threadIdx.x == 0) wg_count = 0;
        __syncthreads();
        open_cnt = wave_sum(open_cnt);
        if (lane == 0 && open_cnt) atomicAdd(&wg_count, open_cnt);
        syncthreads_after_lds_atomics();
        tile_open = wg_count;
    }
    u32 barrier_no = 1;
    if (!pr_grid_barrier(a.ctl, barrier_no++, &bar_flag)) return;
    if (__hip_atomic_load(&a.ctl[PR_CTL_BAIL], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;   // (uniform: read behind the barrier)
    // (from here on the launch goes through: nothing the multi-launch rounds rely on was touched before)
    if (a.lcp_hint) {
#pragma unroll
        for (int j = 0; j < LG_IPT; j++) {
            const u32 local = w * (LG_IPT * WAVE) + j * WAVE + lane;
            if (local < n_act && ((lds.start_bits[local >> 6] >> (local & 63u)) & 1ull)) a.lcp_hint[slot_of[j]] = 0u;
        }
    }

    u32 depth = a.depth;
    // open: the tile holds a group of two or more (tile_open of its elements sit in such groups); dirty: names changed a round ago
    bool wg_open = tile_open > 0u, wg_dirty = false;
    u32 round = 0;
    for (;; round++) {
        const u32 *cur = (round & 1u) ? a.name1 : a.name0;
        u32 *nxt = (round & 1u) ? a.name0 : a.name1;
        // (the thread's position is made opaque once per round: otherwise every mask and LDS address that follows from it
        // -- five rows of them -- is computed in front of the loop and kept, 128 registers and 90 bytes of scratch per lane)
        u32 tid_r = threadIdx.x;
        asm volatile("" : "+v"(tid_r));
        const u32 lane = tid_r & 63u, w = tid_r >> 6;
        PR_STAMP(0);
        bool changed = false;                           // a group of the tile split this round (uniform over the workgroup)
        if (wg_open) {
            // ---- the round keys: the name of the suffix `depth` symbols on ----
            u32 nm[LG_IPT];
            u32 differs = 0;
#pragma unroll
            for (int j = 0; j < LG_IPT; j++) {
                const u32 local = w * (LG_IPT * WAVE) + j * WAVE + lane;
                const u32 q = local < n_act ? local : n_act - 1u;
                // (an element that is alone in its group is final: no name is read for it)
                const bool single = ((lds.start_bits[q >> 6] >> (q & 63u)) & 1ull) && ((lds.start_bits[(q + 1u) >> 6] >> ((q + 1u) & 63u)) & 1ull);
                nm[j] = single || local >= n_act ? 0u : pr_load(&cur[val[j] + depth]);
            }
            // A tile none of whose groups splits has nothing to sort, no bound to move and no name to publish: in the
            // reference's worst case (100 identical strings) a group only splits once the depth reaches its distance from
            // the end of the string -- the early rounds touch a few tiles.  Every member compares with its group's first.
#pragma unroll
            for (int j = 0; j < LG_IPT; j++) {
                const u32 local = w * (LG_IPT * WAVE) + j * WAVE + lane;
                if (local < n_act) lds.vals[local] = nm[j];
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < LG_IPT; j++) {
                const u32 local = w * (LG_IPT * WAVE) + j * WAVE + lane;
                if (local < n_act) differs |= nm[j] ^ lds.vals[pr_group_start(lds.start_bits, local)];
            }
            changed = __syncthreads_or(differs != 0u) != 0;
            PR_STAMP(1);
            if (changed) {
                // ---- number the tile's groups ----
                if (w == 0) {
                    u32 run = 0;
                    for (u32 k = 0; k < LG_WORDS; k += WAVE) {
                        const u32 word = k + lane;
                        u64 x = word < LG_WORDS ? lds.start_bits[word] : 0ull;
                        if (word * 64u >= n_act) x = 0ull;                               // (the padding behind the tile does not count)
                        else if (n_act - word * 64u < 64u) x &= ((u64)1 << (n_act - word * 64u)) - 1ull;
                        const u32 c = (u32)__popcll(x);
                        const u32 inc = wave_inclusive_sum(c);
                        if (word < LG_WORDS) lds.word_prefix[word] = run + inc - c;
                        run += __shfl(inc, 63, WAVE);
                    }
                    if (lane == 0) lds.hdr[0] = run;
                }
                __syncthreads();
                const u32 n_groups = lds.hdr[0];
                const int bits = a.name_bits + (n_groups > 1u ? 32 - (int)__builtin_clz(n_groups - 1u) : 0);
                if (active) {
#pragma unroll
                    for (int j = 0; j < LG_IPT; j++) {
                        const u32 local = w * (LG_IPT * WAVE) + j * WAVE + lane;
                        const u32 q = local < n_act ? local : n_act - 1u;
                        const u32 gid = lds.word_prefix[q >> 6] + (u32)__popcll(lds.start_bits[q >> 6] & (((u64)2 << (q & 63u)) - 1ull)) - 1u;
                        key[j] = local < n_act ? ((u64)gid << a.name_bits) | (u64)nm[j] : ~0ull;
                        if (local >= n_act) val[j] = 0u;
                    }
                }
                for (int shift = 0; shift < bits; shift += 8) lg_radix_pass(lds, key, val, shift, active);
                PR_STAMP(2);
                // ---- the new group bounds, by tile position; where a seam opens, what the two sides are known to share ----
#pragma unroll
                for (int j = 0; j < LG_IPT; j++) {
                    const u32 local = w * (LG_IPT * WAVE) + j * WAVE + lane;
                    const bool st = !active || local >= n_act || local == 0u || key[j] != lds.keys[local - 1u];
                    const bool was = (lds.start_bits[w * LG_IPT + j] >> lane) & 1ull;
                    if (a.lcp_hint && st && !was && local < n_act) a.lcp_hint[slot_of[j]] = depth;
                    const u64 bal = __ballot(st);
                    if (lane == 0) lds.start_bits[w * LG_IPT + j] = bal;
                }
                __syncthreads();
                PR_STAMP(5);
            }
        }
        // ---- publish the names (a tile whose names changed a round ago still owes the other copy) ----
        if (changed || wg_dirty) {
            u32 open_cnt = 0;
            // (only a name that differs from what this copy holds is written -- a write-through store is a fabric write of
            // its own, and most names of a round stay as they are; a tile's own elements are its to read.  All five reads
            // first: one round trip, not five)
            u32 held[LG_IPT];
#pragma unroll
            for (int j = 0; j < LG_IPT; j++) {
                const u32 local = w * (LG_IPT * WAVE) + j * WAVE + lane;
                held[j] = pr_load(&nxt[local < n_act ? val[j] : 0u]);
            }
#ifdef PR_STAMPS
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            PR_STAMP(6);
#endif
#pragma unroll
            for (int j = 0; j < LG_IPT; j++) {
                const u32 local = w * (LG_IPT * WAVE) + j * WAVE + lane;
                if (local >= n_act) continue;
                const u32 name = slot_of[j] - (local - pr_group_start(lds.start_bits, local));
                if (held[j] != name) pr_store(&nxt[val[j]], name);
                const bool single = ((lds.start_bits[local >> 6] >> (local & 63u)) & 1ull) &&
                                    ((lds.start_bits[(local + 1u) >> 6] >> ((local + 1u) & 63u)) & 1ull);
                open_cnt += single ? 0u : 1u;
            }
#ifdef PR_STAMPS
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            PR_STAMP(7);
#endif
            if (threadIdx.x == 0) wg_count = 0;
            __syncthreads();
            open_cnt = wave_sum(open_cnt);
            if (lane == 0 && open_cnt) atomicAdd(&wg_count, open_cnt);
            syncthreads_after_lds_atomics();
            tile_open = wg_count;
        }
        wg_dirty = changed;
        wg_open = tile_open > 0u;
        if (threadIdx.x == 0 && tile_open)
            __hip_atomic_fetch_add(&a.ctl[PR_CTL_OPEN + round], tile_open, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        PR_STAMP(3);
#ifdef PR_STAMPS
        if (round == 3 && threadIdx.x == 0) ((unsigned long long *)(a.ctl + PR_CTL_WORDS))[blockIdx.x * 2] = wall_clock64();
#endif
        if (!pr_grid_barrier(a.ctl, barrier_no++, &bar_flag)) return;
#ifdef PR_STAMPS
        if (round == 3 && threadIdx.x == 0) ((unsigned long long *)(a.ctl + PR_CTL_WORDS))[blockIdx.x * 2 + 1] = wall_clock64();
#endif
        PR_STAMP(4);
        const u32 all_open = __hip_atomic_load(&a.ctl[PR_CTL_OPEN + round], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (all_open == 0u) break;
        if (round + 1u == PR_MAX_ROUNDS) {
            if (threadIdx.x == 0) __hip_atomic_store(&a.ctl[PR_CTL_ABORT], 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        depth *= 2u;
    }
    // ---- every suffix of the tile to its slot ----
#pragma unroll
    for (int j = 0; j < LG_IPT; j++) {
        const u32 local = w * (LG_IPT * WAVE) + j * WAVE + lane;
        if (local < n_act) a.order_g[slot_of[j]] = val[j];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) a.ctl[PR_CTL_ROUNDS] = round + 1u;
}

// ================================================================================================================
// The same rounds for a domain of ANY size: a workgroup walks several tiles per round, and a tile's state -- its
// elements (permuted in place), the group bounds -- lives in global memory between rounds instead of in LDS.  (The
// reference's worst case at n = 10^5 is a domain of 10 M: 4 883 tiles; launch by launch its 15 rounds cost 13 ms.)
// What changes against refine_persist_kernel:
//   * (measured and not kept: ONE barrier per round with double-buffered names -- a tile brings the other copy up to date
//     for its current range when it, or one of the two tiles to its left, changed a round ago --: correct, and no faster,
//     1.07 against 1.05 ms at n = 10^4, 7.1 against 6.8 at n = 10^5: the extra reads of the owed names cost what the barrier
//     saved.)
//   * a round is two phases with a grid barrier behind each.  SORT: every tile reads the names it needs and, where a
//     group splits, sorts, writes its elements back, the suffix array, the LCP hints and the new group bounds.  NAMES:
//     the tiles that changed publish the new names.  One copy of the names is enough: nobody writes a name while
//     anybody may read one.
//   * the group bounds are double-buffered (a round reads one copy of the flags and writes the other): a tile finds its
//     range from the flags -- the groups that start in its chunk --, and the range of its right neighbour begins where a
//     group of its own may have just split; with one copy a neighbour could read that new bound while the elements behind
//     it are still being written.  The ranges of a round partition the domain, so the copy a round writes is complete.
//   * a tile's range may move from round to round (a bound that appears inside chunk c + 1 hands the rest of tile c's
//     last group to tile c + 1); the NAMES phase works on the range its SORT phase had (range[]).
struct Pr2Args {
    u32 *elems;                             // the compacted domain's suffixes (permuted in place)
    const u32 *slots;                       // its slots
    u32 *flags0, *flags1;                   // group start flags: the copy even rounds read (the compaction's) / odd rounds read
    u32 m, depth;
    int name_bits;
    u32 *name;                              // names by text position (every placed suffix initialised)
    u32 *order_g, *lcp_hint, *ctl;
    uint2 *range;                           // per tile: first position (relative to the chunk) and length, as the SORT phase saw them
    u32 *changed;                           // per tile: its groups split in this round's SORT phase
    u32 n_tiles;
};

// The tile of chunk `base / LG_CHUNK` under `gstart`: the groups that start in the chunk (as refine_lds_sort_kernel takes
// them).  All threads call; leaves chunk-relative start bits in lds.start_bits.
// (lane / w: the thread's position, made opaque by the caller once per tile -- PR2_POSITION --: what follows from it would
// otherwise be computed in front of the loops over rounds and tiles and kept there, hundreds of bytes of scratch per lane)
#define PR2_POSITION                              \
    u32 tid_r = threadIdx.x;                      \
    asm volatile("" : "+v"(tid_r));               \
    const u32 lane = tid_r & 63u, w = tid_r >> 6
__device__ __forceinline__ void pr2_tile_bounds(LgLds &lds, const u32 *__restrict__ gstart, u32 m, u32 base, u32 lane, u32 w, u32 &begin_q,
                                                u32 &n_act, bool &too_long)
{
    __syncthreads();                                    // (the tile before is done with lds)
    {
        constexpr int PER = LG_WORDS / LG_WAVES;
        u32 gs[PER];
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const u64 p = (u64)base + (w + (u32)i * LG_WAVES) * 64u + lane;
            gs[i] = pr_load(&gstart[p < m ? p : (u64)m - 1u]);
        }
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const u32 word = w + (u32)i * LG_WAVES;
            const u64 p = (u64)base + word * 64u + lane;
            const bool st = p < m ? gs[i] != 0u : p == m;
            const u64 bal = __ballot(st);
            if (lane == 0) lds.start_bits[word] = bal;
        }
    }
    __syncthreads();
    if (w == 0) {
        const u64 wv = lane < LG_CHUNK / 64 ? lds.start_bits[lane] : 0ull;
        const u64 nzb = __ballot(wv != 0ull);
        u32 bq = LG_NONE, eq = LG_NONE;
        bool tl = false;
        if (nzb) {
            const u32 fl = (u32)__ffsll((unsigned long long)nzb) - 1u, ll = 63u - (u32)__builtin_clzll(nzb);
            bq = fl * 64u + (u32)__builtin_ctzll(lds.start_bits[fl]);
            const u32 last_q = ll * 64u + 63u - (u32)__builtin_clzll(lds.start_bits[ll]);
            const u32 wi = ll + lane;
            u64 x = wi < LG_WORDS ? lds.start_bits[wi] : 0ull;
            if (lane == 0) x &= ~(((u64)2 << (last_q & 63u)) - 1ull);
            const u64 nb = __ballot(x != 0ull);
            u32 next_q = LG_NONE;
            if (nb) {
                const u32 l2 = (u32)__ffsll((unsigned long long)nb) - 1u;
                const u64 xw = ((u64)__shfl((u32)(x >> 32), l2, WAVE) << 32) | __shfl((u32)x, l2, WAVE);
                next_q = (ll + l2) * 64u + (u32)__builtin_ctzll(xw);
            }
            const bool sentinel = (u64)base + last_q == (u64)m;
            tl = !sentinel && (next_q == LG_NONE || next_q - last_q > LG_MAX_GROUP);
            eq = sentinel || tl ? last_q : next_q;
            if (bq == last_q && sentinel) bq = LG_NONE;
        }
        if (lane == 0) {
            lds.hdr[0] = bq;
            lds.hdr[1] = (bq == LG_NONE ? 0u : eq - bq) | (tl ? 0x80000000u : 0u);
        }
    }
    __syncthreads();
    begin_q = lds.hdr[0];
    n_act = lds.hdr[1] & 0x7FFFFFFFu;
    too_long = (lds.hdr[1] >> 31) != 0u;
}

// The tile's elements, slots and group bounds (tile coordinates, all ones behind the tile) for the thread's positions;
// returns the number of the tile's elements that sit in groups of two or more.
__device__ __forceinline__ u32 pr2_tile_load(LgLds &lds, u32 *wg_count, const u32 *__restrict__ flags, const u32 *__restrict__ elems,
                                             const u32 *__restrict__ slots, u32 base, u32 lane, u32 w, u32 begin_q, u32 n_act,
                                             u32 (&val)[LG_IPT], u32 (&slot_of)[LG_IPT])
{
    __syncthreads();                                    // (the chunk-relative bits / the tile before: done with)
#pragma unroll
    for (int j = 0; j < LG_IPT; j++) {
        const u32 local = w * (LG_IPT * WAVE) + j * WAVE + lane;
        const bool in = local < n_act;
        const u64 r = (u64)base + begin_q + (in ? local : 0u);
        val[j] = in ? pr_load(&elems[r]) : 0u;
        slot_of[j] = in ? slots[r] : 0u;
        const u32 st = in ? pr_load(&flags[r]) : 1u;
        const u64 bal = __ballot(st != 0u);
        if (lane == 0) lds.start_bits[w * LG_IPT + j] = bal;
    }
    if (threadIdx.x == 0) { lds.start_bits[LG_WORDS] = ~0ull; *wg_count = 0; }
    __syncthreads();
    u32 open_cnt = 0;
#pragma unroll
    for (int j = 0; j < LG_IPT; j++) {
        const u32 local = w * (LG_IPT * WAVE) + j * WAVE + lane;
        if (local >= n_act) continue;
        const bool single = ((lds.start_bits[local >> 6] >> (local & 63u)) & 1ull) && ((lds.start_bits[(local + 1u) >> 6] >> ((local + 1u) & 63u)) & 1ull);
        open_cnt += single ? 0u : 1u;
    }
    open_cnt = wave_sum(open_cnt);
    if (lane == 0 && open_cnt) atomicAdd(wg_count, open_cnt);
    syncthreads_after_lds_atomics();
    return *wg_count;
}

// (one workgroup per CU: two of them share a CU's SIMDs without gaining on each other, and at 128 registers a lane nothing
// is spilled -- measured on this kernel too: at two per CU, 64 registers, 2.25 against 1.04 ms at n = 10^4, 11.4 against 6.9 at 10^5)
__global__ __launch_bounds__(LG_THREADS, 4) void refine_persist2_kernel(Pr2Args a)
{
    __shared__ LgLds lds;
    __shared__ u32 bar_flag, wg_count;
    const u32 m = a.m;
    u32 barrier_no = 1;
    // ---- can every group be sorted inside a tile?  (nothing is changed before that is known) ----
    for (u32 t = blockIdx.x; t < a.n_tiles; t += gridDim.x) {
        PR2_POSITION;
        u32 begin_q, n_act;
        bool too_long;
        pr2_tile_bounds(lds, a.flags0, m, t * LG_CHUNK, lane, w, begin_q, n_act, too_long);
        if (too_long && threadIdx.x == 0) __hip_atomic_store(&a.ctl[PR_CTL_BAIL], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!pr_grid_barrier(a.ctl, barrier_no++, &bar_flag)) return;
    if (__hip_atomic_load(&a.ctl[PR_CTL_BAIL], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    // ---- every member is named by the slot where its group starts; the hints of the first ranks of groups ----
    for (u32 t = blockIdx.x; t < a.n_tiles; t += gridDim.x) {
        PR2_POSITION;
        const u32 base = t * LG_CHUNK;
        u32 begin_q, n_act, val[LG_IPT], slot_of[LG_IPT];
        bool too_long;
        pr2_tile_bounds(lds, a.flags0, m, base, lane, w, begin_q, n_act, too_long);
        if (n_act == 0) continue;
        (void)pr2_tile_load(lds, &wg_count, a.flags0, a.elems, a.slots, base, lane, w, begin_q, n_act, val, slot_of);
#pragma unroll
        for (int j = 0; j < LG_IPT; j++) {
            const u32 local = w * (LG_IPT * WAVE) + j * WAVE + lane;
            if (local >= n_act) continue;
            const u32 nm = slot_of[j] - (local - pr_group_start(lds.start_bits, local));
            pr_store(&a.name[val[j]], nm);
            a.order_g[slot_of[j]] = val[j];                 // (a tile that never changes again has its suffixes in place)
            if (a.lcp_hint && nm == slot_of[j]) a.lcp_hint[nm] = 0u;
        }
    }
    if (!pr_grid_barrier(a.ctl, barrier_no++, &bar_flag)) return;

    u32 depth = a.depth, round = 0;
    for (;; round++) {
        const u32 *f_cur = (round & 1u) ? a.flags1 : a.flags0;
        u32 *f_next = (round & 1u) ? a.flags0 : a.flags1;
        u32 my_open = 0;
        // ---- SORT: every tile of this workgroup ----
        for (u32 t = blockIdx.x; t < a.n_tiles; t += gridDim.x) {
            PR2_POSITION;
            const u32 base = t * LG_CHUNK;
            u32 begin_q, n_act, val[LG_IPT], slot_of[LG_IPT];
            bool too_long;
            pr2_tile_bounds(lds, f_cur, m, base, lane, w, begin_q, n_act, too_long);
            // (this workgroup's own notes -- but read rounds later through an L1 that may still hold the line of a round before:
            // written through and read past it like everything else a round leaves behind)
            if (threadIdx.x == 0) { pr_store(&a.range[t].x, begin_q); pr_store(&a.range[t].y, n_act); pr_store(&a.changed[t], 0u); }
            if (n_act == 0) continue;
            const bool active = w * (LG_IPT * WAVE) < n_act;
            const u32 tile_open = pr2_tile_load(lds, &wg_count, f_cur, a.elems, a.slots, base, lane, w, begin_q, n_act, val, slot_of);
            bool changed = false;
            u32 nm[LG_IPT];
            if (tile_open) {
                u32 differs = 0;
#pragma unroll
                for (int j = 0; j < LG_IPT; j++) {
                    const u32 local = w * (LG_IPT * WAVE) + j * WAVE + lane;
                    const u32 q = local < n_act ? local : n_act - 1u;
                    const bool single = ((lds.start_bits[q >> 6] >> (q & 63u)) & 1ull) && ((lds.start_bits[(q + 1u) >> 6] >> ((q + 1u) & 63u)) & 1ull);
                    nm[j] = single || local >= n_act ? 0u : pr_load(&a.name[val[j] + depth]);
                }
#pragma unroll
                for (int j = 0; j < LG_IPT; j++) {
                    const u32 local = w * (LG_IPT * WAVE) + j * WAVE + lane;
                    if (local < n_act) lds.vals[local] = nm[j];
                }
                __syncthreads();
#pragma unroll
                for (int j = 0; j < LG_IPT; j++) {
                    const u32 local = w * (LG_IPT * WAVE) + j * WAVE + lane;
                    if (local < n_act) differs |= nm[j] ^ lds.vals[pr_group_start(lds.start_bits, local)];
                }
                changed = __syncthreads_or(differs != 0u) != 0;
            }
            if (!changed) {
                // nothing moves: the bounds are copied into the other copy of the flags
#pragma unroll
                for (int j = 0; j < LG_IPT; j++) {
                    const u32 local = w * (LG_IPT * WAVE) + j * WAVE + lane;
                    if (local < n_act) pr_store(&f_next[(u64)base + begin_q + local], (u32)((lds.start_bits[local >> 6] >> (local & 63u)) & 1ull));
                }
                my_open += tile_open;
                continue;
            }
            // ---- number the tile's groups, sort by (group, name) ----
            if (w == 0) {
                u32 run = 0;
                for (u32 k = 0; k < LG_WORDS; k += WAVE) {
                    const u32 word = k + lane;
                    u64 x = word < LG_WORDS ? lds.start_bits[word] : 0ull;
                    if (word * 64u >= n_act) x = 0ull;
                    else if (n_act - word * 64u < 64u) x &= ((u64)1 << (n_act - word * 64u)) - 1ull;
                    const u32 c = (u32)__popcll(x);
                    const u32 inc = wave_inclusive_sum(c);
                    if (word < LG_WORDS) lds.word_prefix[word] = run + inc - c;
                    run += __shfl(inc, 63, WAVE);
                }
                if (lane == 0) lds.hdr[0] = run;
            }
            __syncthreads();
            const u32 n_groups = lds.hdr[0];
            const int bits = a.name_bits + (n_groups > 1u ? 32 - (int)__builtin_clz(n_groups - 1u) : 0);
            u64 key[LG_IPT];
            if (active) {
#pragma unroll
                for (int j = 0; j < LG_IPT; j++) {
                    const u32 local = w * (LG_IPT * WAVE) + j * WAVE + lane;
                    const u32 q = local < n_act ? local : n_act - 1u;
                    const u32 gid = lds.word_prefix[q >> 6] + (u32)__popcll(lds.start_bits[q >> 6] & (((u64)2 << (q & 63u)) - 1ull)) - 1u;
                    key[j] = local < n_act ? ((u64)gid << a.name_bits) | (u64)nm[j] : ~0ull;
                    if (local >= n_act) val[j] = 0u;
                }
            }
            for (int shift = 0; shift < bits; shift += 8) lg_radix_pass(lds, key, val, shift, active);
            // ---- the new bounds (into the other copy of the flags), the elements back, the suffix array, the hints ----
#pragma unroll
            for (int j = 0; j < LG_IPT; j++) {
                const u32 local = w * (LG_IPT * WAVE) + j * WAVE + lane;
                const bool st = !active || local >= n_act || local == 0u || key[j] != lds.keys[local - 1u];
                const bool was = (lds.start_bits[w * LG_IPT + j] >> lane) & 1ull;
                if (local < n_act) {
                    const u64 r = (u64)base + begin_q + local;
                    pr_store(&f_next[r], st ? 1u : 0u);
                    pr_store(&a.elems[r], val[j]);
                    a.order_g[slot_of[j]] = val[j];
                    if (a.lcp_hint && st && !was) a.lcp_hint[slot_of[j]] = depth;
                }
                const u64 bal = __ballot(st);
                if (lane == 0) lds.start_bits[w * LG_IPT + j] = bal;
            }
            if (threadIdx.x == 0) { pr_store(&a.changed[t], 1u); wg_count = 0; }
            __syncthreads();
            u32 open_cnt = 0;
#pragma unroll
            for (int j = 0; j < LG_IPT; j++) {
                const u32 local = w * (LG_IPT * WAVE) + j * WAVE + lane;
                if (local >= n_act) continue;
                const bool single = ((lds.start_bits[local >> 6] >> (local & 63u)) & 1ull) && ((lds.start_bits[(local + 1u) >> 6] >> ((local + 1u) & 63u)) & 1ull);
                open_cnt += single ? 0u : 1u;
            }
            open_cnt = wave_sum(open_cnt);
            if (lane == 0 && open_cnt) atomicAdd(&wg_count, open_cnt);
            syncthreads_after_lds_atomics();
            my_open += wg_count;
        }
        if (threadIdx.x == 0 && my_open)
            __hip_atomic_fetch_add(&a.ctl[PR_CTL_OPEN + round], my_open, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!pr_grid_barrier(a.ctl, barrier_no++, &bar_flag)) return;
        const u32 all_open = __hip_atomic_load(&a.ctl[PR_CTL_OPEN + round], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // ---- NAMES: the tiles that changed publish the new names (the last round's are of no use to anybody) ----
        if (all_open) {
            for (u32 t = blockIdx.x; t < a.n_tiles; t += gridDim.x) {
                if (!pr_load(&a.changed[t])) continue;      // (this workgroup's own write: uniform)
                PR2_POSITION;
                const u32 base = t * LG_CHUNK;
                const uint2 rg = uint2{pr_load(&a.range[t].x), pr_load(&a.range[t].y)};
                u32 val[LG_IPT], slot_of[LG_IPT], held[LG_IPT];
                (void)pr2_tile_load(lds, &wg_count, f_next, a.elems, a.slots, base, lane, w, rg.x, rg.y, val, slot_of);
#pragma unroll
                for (int j = 0; j < LG_IPT; j++) {
                    const u32 local = w * (LG_IPT * WAVE) + j * WAVE + lane;
                    held[j] = pr_load(&a.name[local < rg.y ? val[j] : 0u]);
                }
#pragma unroll
                for (int j = 0; j < LG_IPT; j++) {
                    const u32 local = w * (LG_IPT * WAVE) + j * WAVE + lane;
                    if (local >= rg.y) continue;
                    const u32 name = slot_of[j] - (local - pr_group_start(lds.start_bits, local));
                    if (held[j] != name) pr_store(&a.name[val[j]], name);
                }
            }
        }
        if (!pr_grid_barrier(a.ctl, barrier_no++, &bar_flag)) return;
        if (all_open == 0u) break;
        if (round + 1u == PR_MAX_ROUNDS) {
            if (threadIdx.x == 0) __hip_atomic_store(&a.ctl[PR_CTL_ABORT], 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        depth *= 2u;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) a.ctl[PR_CTL_ROUNDS] = round + 1u;
}
