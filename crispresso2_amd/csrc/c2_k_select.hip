// c2_k_select.hip -- the seed test (strand plan) and the strand / best-reference choice of get_new_variant_object on the device.
#pragma once
#include "c2_k_common.h"

// =====================================================================================
// The seed test of get_new_variant_object (CRISPRessoCORE.py:656-687) for a batch of reads that are already on the device:
// found_fw / found_rc = how many of the reference's first n seeds (forward / reverse complement) occur in the read (Python `in`:
// the empty seed always does, a seed longer than the read never); plan = 0 forward only (found_fw > seed_min and found_rc == 0),
// 1 reverse complement only (found_fw == 0 and found_rc > seed_min), else 2 (both strands are aligned).  One wavefront per read:
// the read goes to LDS, lane p tests the window that starts at p (+64, +128, ...), a ballot says whether any window matched.
// Same answers as the host's c2_strand_plan (tests/test_select_emulated.py, test_gpu_parity.py).
// =====================================================================================
__global__ __launch_bounds__(256) void c2_strand_plan_kernel(c2_strand_args A)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t row = c2_strand_row_bytes(A.max_read_len);
    unsigned char* sRead = c2_smem + (uint32_t)wave * row;
    unsigned char* sSeeds = c2_smem + 4u * row;                      // [n_refs][2][max_seeds][C2_SEED_SLOT], zero padded (seed_table)
    if (A.seed_table) {
        const int n_slots = A.n_refs * 2 * A.max_seeds;
        for (int e = threadIdx.x; e < n_slots * (int)(C2_SEED_SLOT / 4u); e += blockDim.x) ((uint32_t*)sSeeds)[e] = 0u;
        __syncthreads();
        for (int e = threadIdx.x; e < n_slots * (int)C2_SEED_SLOT; e += blockDim.x) {
            const int slot = e / (int)C2_SEED_SLOT, k = e % (int)C2_SEED_SLOT;
            if (k < A.seed_len[slot]) sSeeds[e] = A.seed_blob[A.seed_off[slot] + k];
        }
        __syncthreads();
    }
    for (uint64_t i = (uint64_t)blockIdx.x * 4u + (uint64_t)wave; i < A.n_reads; i += (uint64_t)gridDim.x * 4u) {
        const uint64_t o = A.offsets[i];
        const int Lj = (int)(A.offsets[i + 1] - o);
        for (int k = lane; k < Lj; k += 64) sRead[k] = A.reads[o + (uint64_t)k];
        __builtin_amdgcn_wave_barrier();
        for (int r = 0; r < A.n_refs; ++r) {
            int found[2] = {0, 0};
            const int ns = A.n_seeds[r];
            for (int st = 0; st < 2; ++st)
                for (int q = 0; q < ns; ++q) {
                    const int idx = (r * 2 + st) * A.max_seeds + q;
                    const int len = A.seed_len[idx];
                    if (len == 0) { ++found[st]; continue; }
                    if (len > Lj) continue;
                    bool any = false;
                    if (A.seed_table) {
                        // four bytes of the seed against four bytes of the window at a time: the window dword at byte p + 4 j comes out of the two
                        // aligned LDS dwords around it (v_alignbyte); the seed's dwords are read once (same address in every lane: a broadcast)
                        const uint32_t* sd = (const uint32_t*)(sSeeds + (uint32_t)idx * C2_SEED_SLOT);
                        const int nd = (len + 3) >> 2;
                        const uint32_t last_mask = (len & 3) ? ((1u << (8 * (len & 3))) - 1u) : 0xffffffffu;
                        const uint32_t* sRead32 = (const uint32_t*)sRead;
                        for (int base = 0; base + len <= Lj && !any; base += 64) {
                            const int p = base + lane;
                            const bool in = p + len <= Lj;
                            const int pc = in ? p : 0;                 // (a lane without a window reads the first one: inside the row)
                            uint32_t diff = 0;
                            for (int j = 0; j < nd; ++j) {
                                const int b = pc + 4 * j;
                                const uint32_t lo = sRead32[b >> 2], hi = sRead32[(b >> 2) + 1];
                                const uint32_t w = __builtin_amdgcn_alignbyte(hi, lo, (unsigned)(b & 3));
                                diff |= (w ^ sd[j]) & (j == nd - 1 ? last_mask : 0xffffffffu);
                            }
                            any = __ballot(in && diff == 0u) != 0ull;
                        }
                    } else {
                        const uint8_t* seed = A.seed_blob + A.seed_off[idx];
                        for (int base = 0; base + len <= Lj && !any; base += 64) {
                            const int p = base + lane;
                            bool ok = p + len <= Lj;
                            for (int k = 0; k < len && __ballot(ok) != 0ull; ++k) ok = ok && sRead[ok ? p + k : 0] == seed[k];
                            any = __ballot(ok) != 0ull;
                        }
                    }
                    if (any) ++found[st];
                }
            if (lane == 0)
                A.plan[i * (uint64_t)A.n_refs + (uint64_t)r] = (found[0] > A.seed_min && found[1] == 0) ? 0 : (found[0] == 0 && found[1] > A.seed_min) ? 1 : 2;
        }
        __builtin_amdgcn_wave_barrier();                            // (the next read overwrites the row)
    }
}

// =====================================================================================
// Strand and best-reference choice of get_new_variant_object on the device (CRISPRessoCORE.py:683 strand: strict '>';
// :697-707 best reference: first strictly better score that also exceeds refs[name]['min_aln_score'], later equal scores
// join; :710 aligned iff the best score is > 0; :779-785 ambiguous reads), one lane per read over its k records.
// The reference compares Python floats round(100*matches/float(len), 3); here the same order on integers: c2_mscore is
// 1000 x that rounded value.  100000*m/T is a multiple of 1/T, so unless it is an exact tie it lies >= 1/(2T) from the
// rounding boundary -- far more than the double's error -- and an exact tie has the form odd/2000 = x with T a multiple
// of 64 * 5^j (T < 8000): x is then a dyadic rational, the double is exact and Python rounds half to even.  The host
// refuses alignments of 8000 columns and more (c2_select_best_device).
// =====================================================================================
__host__ __device__ inline uint32_t c2_mscore(const uint32_t matches, const uint32_t T) {
    if (T == 0) return 0;
    const uint64_t num = 100000ull * matches;                        // < 2^33
    // floor(num / T) through one double division (exact operands; the quotient may be off by one): corrected with integers
    int64_t q = (int64_t)((double)num / (double)T);
    int64_t r = (int64_t)num - q * (int64_t)T;
    if (r < 0) { --q; r += T; } else if (r >= (int64_t)T) { ++q; r -= T; }
    if (2 * r > (int64_t)T) ++q; else if (2 * r == (int64_t)T) q += (q & 1);
    return (uint32_t)q;
}

__global__ __launch_bounds__(256) void c2_select_best_kernel(c2_select_args A)
{
    const uint64_t read = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    unsigned long long st[C2_SEL_STATS];
#pragma unroll
    for (int q = 0; q < C2_SEL_STATS; ++q) st[q] = 0;
    if (read < A.n_reads) {
        const int k = A.n_refs;
        const int W = (k + 63) >> 6;                                         // 64-bit words of a read's masks (bit r of word r / 64: reference r)
        // the score the choice compares for (read, reference r), and whether it is the reverse-complement batch's (:683)
        auto score_of = [&](const int r, bool& second, const bool tally) -> long long {
            const uint64_t t = read * (uint64_t)k + (uint64_t)r;
            const c2_aln_record* rec = A.records + t;
            // (what the choice needs of a record -- length, matches, status -- with two loads instead of one per field)
            const unsigned w0 = ((const unsigned*)rec)[0], w5 = ((const unsigned*)rec)[5];
            const unsigned rstatus = w5 >> 24;
            if (tally && rstatus != 0) { st[C2_SEL_N_BAD_STATUS] += 1; st[C2_SEL_FIRST_BAD_STATUS] = rstatus; }
            long long ms = (long long)c2_mscore(w0 >> 16, w0 & 0xffffu);
            second = false;
            if (A.records2 && A.slot2) {
                const int sl = A.slot2[t];
                if (sl >= 0) {
                    const c2_aln_record* rec2 = A.records2 + sl;
                    if (tally && rec2->status != 0) { st[C2_SEL_N_BAD_STATUS] += 1; st[C2_SEL_FIRST_BAD_STATUS] = rec2->status; }
                    const long long ms2 = (long long)c2_mscore(rec2->matches, rec2->aln_len);
                    if (ms2 > ms) { ms = ms2; second = true; }              // :683  if (rvscore > fwscore)
                }
            }
            return ms;
        };
        // pass 1: the best score, where it was set, how many references share it, the last of them.  The reference's loop
        // (:697 `if score > best_match_score and score > min_aln_score: best_match_names = [name]`, :703 `elif score == best_match_score:
        // append`) makes reference r a best match iff its score equals the final best AND r is not in front of the reference that set
        // it (one in front with that score did not pass its threshold, or it would have set the best itself).
        long long best = -1;
        int first = -1, last = -1, nb = 0; bool last2 = false;
        for (int r = 0; r < k; ++r) {
            bool second;
            const long long ms = score_of(r, second, true);
            if (ms > best && ms >= (long long)A.min_mscore[r]) { best = ms; first = r; nb = 1; last = r; last2 = second; }   // :697
            else if (ms == best) { ++nb; last = r; last2 = second; }                                                           // :703
        }
        const bool aligned = best > 0;                                       // :710
        const bool ambiguous = aligned && nb > 1 && A.mode != C2_SEL_MODE_FIRST && A.mode != C2_SEL_MODE_EXPAND;
        if (A.flags) A.flags[read] = (uint8_t)((aligned ? C2_SEL_FLAG_ALIGNED : 0) | (ambiguous ? C2_SEL_FLAG_AMBIGUOUS : 0));
        // pass 2: the masks, a word at a time, and the weight of every alignment in the count pass
        if (A.member || A.use2 || A.weights) {
            const uint32_t w = A.counts ? A.counts[read] : 1u;
            unsigned long long member = 0, use2 = 0;
            for (int r = 0; r < k; ++r) {
                bool second;
                const long long ms = score_of(r, second, false);
                const bool m = aligned && r >= first && ms == best;
                // counted: every best match (one of them, or --expand_ambiguous_alignments), the first one
                // (--assign_ambiguous_alignments_to_first_reference), none when the read is ambiguous
                const bool c = m && (nb == 1 || A.mode == C2_SEL_MODE_EXPAND || (A.mode == C2_SEL_MODE_FIRST && r == first));
                if (m) member |= 1ull << (r & 63);
                if (second) use2 |= 1ull << (r & 63);
                if (A.weights) {
                    const uint64_t t = read * (uint64_t)k + (uint64_t)r;
                    A.weights[t] = (c && !second) ? w : 0u;
                    if (A.weights2 && A.slot2 && A.slot2[t] >= 0) A.weights2[A.slot2[t]] = (c && second) ? w : 0u;
                }
                if ((r & 63) == 63 || r == k - 1) {
                    if (A.member) A.member[read * (uint64_t)W + (uint64_t)(r >> 6)] = member;
                    if (A.use2) A.use2[read * (uint64_t)W + (uint64_t)(r >> 6)] = use2;
                    member = 0; use2 = 0;
                }
            }
        }
        // aln_stats of process_fastq (:1974-1979): payload of the LAST best match, raw multiplicity
        const unsigned long long raw = A.raw_counts ? (unsigned long long)A.raw_counts[read] : 1ull;
        if (aligned) {
            st[C2_SEL_N_COMPUTED_ALN] = 1; st[C2_SEL_N_CACHED_ALN] = raw - 1ull;
            const c2_aln_record* rec = (last2 ? A.records2 + A.slot2[read * (uint64_t)k + (uint64_t)last] : A.records + read * (uint64_t)k + (uint64_t)last);
            const unsigned long long sub_all = rec->all_substitutions, sub_win = rec->substitution_n;
            const unsigned long long total = (unsigned long long)rec->all_insertion_events + rec->all_deletion_bases + sub_all;
            const unsigned long long in_win = sub_win + rec->deletion_n + rec->insertion_n;
            st[C2_SEL_N_GLOBAL_SUBS] = sub_all * raw;
            st[C2_SEL_N_SUBS_OUTSIDE_WINDOW] = (sub_all - sub_win) * raw;
            st[C2_SEL_N_MODS_IN_WINDOW] = in_win * raw;
            st[C2_SEL_N_MODS_OUTSIDE_WINDOW] = (total - in_win) * raw;       // (two's complement like the host's int64 sum)
            st[C2_SEL_N_READS_IRREGULAR_ENDS] = (unsigned long long)rec->irregular_ends * raw;
        } else {
            st[C2_SEL_N_COMPUTED_NOTALN] = 1; st[C2_SEL_N_CACHED_NOTALN] = raw - 1ull;
        }
    }
    if (A.stats) {
        // block sums in LDS (a thread adds only what is non-zero: a handful of LDS atomics), then one global atomic per statistic
        // per block; FIRST_BAD_STATUS: any non-zero value will do (plain store)
        unsigned long long* blk = (unsigned long long*)c2_smem;       // (launched with C2_SEL_STATS * 8 bytes of dynamic LDS)
        if (threadIdx.x < C2_SEL_STATS) blk[threadIdx.x] = 0ull;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < C2_SEL_STATS; ++q) {
            if (q == C2_SEL_FIRST_BAD_STATUS) { if (st[q] != 0) A.stats[q] = st[q]; continue; }
            if (st[q] != 0) atomicAdd(&blk[q], st[q]);
        }
        __syncthreads();
        if (threadIdx.x < C2_SEL_STATS && threadIdx.x != C2_SEL_FIRST_BAD_STATUS && blk[threadIdx.x] != 0) atomicAdd(A.stats + threadIdx.x, blk[threadIdx.x]);
    }
}
