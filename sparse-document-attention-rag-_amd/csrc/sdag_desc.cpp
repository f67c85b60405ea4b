// Host-side builder of the block-sparse attention descriptor (see common.h for the word layout).
//
// The reference fills an L x L boolean tensor row by row (src/pipeline/sparse_attention_RAG/SDAG.py:104-127).
// Here the same relation is produced 32 rows at a time as one 64-bit word per (row, key tile) and immediately
// reduced to: skipped tiles, fully allowed tiles, and partially allowed tiles that keep their 32 x 64 bit mask.
#include <algorithm>
#include <cstring>
#include <functional>
#include <unordered_map>
#include <vector>

#include "common.h"

namespace sdag {

static thread_local char g_err[512] = "";
char* err_buf() { return g_err; }
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

namespace {

constexpr int RB = SDAG_ROW_BLOCK;
constexpr int KT = SDAG_KV_TILE;
static_assert(RB == 32 && KT == 64, "descriptor masks are 32 rows of one 64-bit word");

// columns [a, b), b > a, of one bit row; `touched[t]` is set for every 64-bit word t written
inline void set_range(uint64_t* row_bits, uint8_t* touched, int a, int b) {
    const int t0 = a >> 6, t1 = (b - 1) >> 6;
    for (int t = t0; t <= t1; ++t) {
        const int lo = t == t0 ? (a & 63) : 0;
        const int hi = t == t1 ? ((b - 1) & 63) : 63;  // inclusive
        const uint64_t m = (hi == 63 ? ~0ull : ((1ull << (hi + 1)) - 1)) & ~((1ull << lo) - 1);
        row_bits[t] |= m;
        touched[t] = 1;
    }
}

// One unit as the scheduler sees it.
struct SchedUnit {
    int32_t b, row0, item_begin, item_end, slot, ubits, L;
    const int32_t* items;  // the sequence's {first key, mask index} pairs
    int64_t lmask_base;    // word offset (from the start of the table the steps live in) of that sequence's lane masks
    const int32_t* lmask_at = nullptr;  // or: word offset of every lane mask of the sequence (a pool shared by a batch)
};

// Cost of a unit in quarter steps: loading its Q rows and storing its O rows takes about as long as 1.25 tile steps
// (measured: 1.1 to 1.3).
inline int64_t unit_cost(const SchedUnit& u) { return 4 * (int64_t)(u.item_end - u.item_begin) + 5; }
inline bool unit_is_long(const SchedUnit& u) { return u.item_end - u.item_begin >= 14; }

// Deal the units, in the order given, to n_streams streams: each goes to the stream with the least work so far
// (cost = key tiles + a fixed charge per unit for loading Q and storing O); with `half` given (one entry per unit,
// 0 or 1), only among the first or the second half of the streams.  Writes the stream offsets (in steps) and the
// step records.
void build_schedule(const std::vector<SchedUnit>& units, int n_streams, int32_t* stream_off, int32_t* steps,
                    const std::vector<uint8_t>* half = nullptr) {
    std::vector<int64_t> load((size_t)n_streams, 0);
    std::vector<std::vector<int32_t>> mine((size_t)n_streams);
    for (size_t u = 0; u < units.size(); ++u) {
        const int lo = half ? ((*half)[u] ? n_streams / 2 : 0) : 0;
        const int hi = half ? ((*half)[u] ? n_streams : n_streams / 2) : n_streams;
        int best = lo;
        for (int s = lo + 1; s < hi; ++s)
            if (load[s] < load[best]) best = s;
        mine[best].push_back((int32_t)u);
        load[best] += unit_cost(units[u]);
    }
    // Dealt in an order that is not longest-first (the batch order above), the streams end up to one long unit
    // apart.  Level each group of streams: move a unit from the most loaded stream to the least loaded one while
    // that lowers the larger of the two, and keep every stream's units in dealing order.
    if (half)
        for (int g = 0; g < 2; ++g) {
            const int lo = g ? n_streams / 2 : 0, hi = g ? n_streams : n_streams / 2;
            for (int64_t round = 0; round < (int64_t)units.size(); ++round) {
                int smax = lo, smin = lo;
                for (int s = lo + 1; s < hi; ++s) {
                    if (load[s] > load[smax]) smax = s;
                    if (load[s] < load[smin]) smin = s;
                }
                const int64_t gap = load[smax] - load[smin];
                // the unit whose cost is nearest to half the gap, among those smaller than the gap; the latest such
                int pick = -1;
                int64_t pick_err = 0;
                for (int i = (int)mine[smax].size() - 1; i >= 0; --i) {
                    const int64_t c = unit_cost(units[mine[smax][i]]);
                    if (c >= gap) continue;
                    const int64_t err = 2 * c > gap ? 2 * c - gap : gap - 2 * c;
                    if (pick < 0 || err < pick_err) { pick = i; pick_err = err; }
                }
                if (pick < 0) break;
                const int32_t ui = mine[smax][pick];
                const int64_t c = unit_cost(units[ui]);
                mine[smax].erase(mine[smax].begin() + pick);
                mine[smin].insert(std::upper_bound(mine[smin].begin(), mine[smin].end(), ui), ui);
                load[smax] -= c;
                load[smin] += c;
            }
        }
    int64_t n = 0;
    for (int s = 0; s < n_streams; ++s) {
        stream_off[s] = (int32_t)n;
        for (int32_t ui : mine[s]) {
            const SchedUnit& u = units[ui];
            for (int it = u.item_begin; it < u.item_end; ++it, ++n) {
                int32_t* w = steps + n * STEP_WORDS;
                const int32_t mi = u.items[2 * it + 1];
                w[S_KEY0] = u.items[2 * it];
                w[S_MOFF] = mi < 0 ? 0 : u.lmask_at ? u.lmask_at[mi] : (int32_t)(u.lmask_base + (int64_t)mi * 64);
                w[S_FLAGS] = (it == u.item_begin ? 1 : 0) | (it + 1 == u.item_end ? 2 : 0);
                w[S_B] = u.b;
                w[S_L] = u.L;
                w[S_ROW0] = u.row0;
                w[S_SLOT] = u.slot;
                w[S_UBITS] = u.ubits;
            }
        }
    }
    stream_off[n_streams] = (int32_t)n;
}

// In-place transpose of a 32 x 32 bit matrix: afterwards bit r of a[c] is what bit c of a[r] was (five rounds of
// block swaps, 16 x 16 down to 1 x 1).
inline void transpose32(uint32_t* a) {
    uint32_t m = 0x0000ffffu;
    for (int j = 16; j != 0; j >>= 1, m ^= m << j)
        for (int k = 0; k < 32; k = (k + j + 1) & ~j) {
            const uint32_t t = ((a[k] >> j) ^ a[k + j]) & m;
            a[k] ^= t << j;
            a[k + j] ^= t;
        }
}

// 32 rows x 64 bits regrouped per accumulator register of the attention kernel (see common.h): word 16 kb + i holds,
// for h = 0, 1, bit (r + 32 h) = row r's bit of key 32 kb + (i & 3) + 8 (i >> 2) + 4 h -- the columns of the two
// 32 x 32 halves of the mask, paired.
void lane_mask_image(const uint64_t* rows, uint64_t* out) {
    for (int kb = 0; kb < 2; ++kb) {
        uint32_t t[32];
        for (int r = 0; r < 32; ++r) t[r] = (uint32_t)(rows[r] >> (32 * kb));
        transpose32(t);  // t[key] bit r
        for (int i = 0; i < 16; ++i) {
            const int key = (i & 3) + 8 * (i >> 2);
            out[16 * kb + i] = (uint64_t)t[key] | ((uint64_t)t[key + 4] << 32);
        }
    }
}

inline int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

// fill(row0, rows, bits, touched): the allowed keys of rows row0 .. row0 + rows - 1 as bit rows of nw = ceil(L / 64)
// words, bits[r * nw + w], zeroed by the caller; touched[w] = 1 for every word index w it wrote in any row (a row of a
// document sees a few dozen of a long prompt's words: everything after the fill walks those words only).
using FillFn = std::function<void(int, int, uint64_t*, uint8_t*)>;

// 64 bits of a bit row starting at bit `pos` (bits at and beyond 64 * nw read as zero).
inline uint64_t window(const uint64_t* row, int nw, int pos) {
    const int w = pos >> 6, sh = pos & 63;
    uint64_t x = w < nw ? row[w] >> sh : 0;
    if (sh && w + 1 < nw) x |= row[w + 1] << (64 - sh);
    return x;
}

// `cuts`: positions where a row block must start (any order, duplicates and out-of-range values allowed).
int emit(int L, std::vector<int> cuts, const FillFn& fill, int max_items_per_unit, int n_streams, int32_t* out,
         int64_t cap_words, int64_t* n_words) {
    if (n_streams <= 0) n_streams = DEFAULT_STREAMS;
    if (n_streams > 65536) return fail(SDAG_ERR_ARG, "too many streams");
    const int nw = (L + KT - 1) / KT;
    // ---- row blocks
    cuts.push_back(0);
    cuts.push_back(L);
    std::sort(cuts.begin(), cuts.end());
    std::vector<int32_t> rb_row0;
    for (size_t i = 0; i + 1 < cuts.size(); ++i) {
        const int a = cuts[i], b = cuts[i + 1];
        if (a < 0 || b > L || b <= a) continue;
        for (int r = a; r < b; r += RB) rb_row0.push_back(r);
    }
    const int nrb = (int)rb_row0.size();
    auto rb_rows = [&](int rb) { return (rb + 1 < nrb ? rb_row0[rb + 1] : L) - rb_row0[rb]; };

    // uni_cols / all_cols: per 64-bit word of the block's rows, the keys SOME row / EVERY row may attend
    std::vector<uint64_t> bits((size_t)RB * std::max(nw, 1), 0ull), uni_cols((size_t)std::max(nw, 1), 0ull),
        all_cols((size_t)std::max(nw, 1), 0ull);
    std::vector<uint8_t> touched((size_t)std::max(nw, 1), 0);
    std::vector<int32_t> tw;               // the touched word indices of the current block
    std::vector<int32_t> items;            // {key0, mask}
    std::vector<int32_t> rb_item_begin(nrb + 1, 0);
    std::vector<uint64_t> masks;           // 32 words per mask; equal masks are stored once
    // Tiles follow the documents, so the staircase of one document's triangle repeats in every other document: a prompt
    // of ten like documents has a tenth of the distinct masks (less to build, to upload and to pull through the GPU's
    // scalar cache).  hash of a mask -> index of the first mask with that hash, chained through `next_same`.
    std::unordered_map<uint64_t, int32_t> mask_of;
    std::vector<int32_t> next_same;
    std::vector<int32_t> rbinfo((size_t)nrb * 2, 0);
    std::vector<int32_t> uniform;
    int64_t pairs = 0;

    for (int rb = 0; rb < nrb; ++rb) {
        const int row0 = rb_row0[rb], rows = rb_rows(rb);
        // (bits, uni_cols and touched are all zero here: the previous block cleared what it had written)
        fill(row0, rows, bits.data(), touched.data());
        tw.clear();
        for (int t = 0; t < nw; ++t)
            if (touched[t]) tw.push_back(t);
        uint32_t skip = rows < RB ? ~0u << rows : 0u;
        for (int r = 0; r < rows; ++r) {
            uint64_t any = 0;
            const uint64_t* row = bits.data() + (size_t)r * nw;
            for (int32_t t : tw) {
                const uint64_t w = row[t];
                any |= w;
                uni_cols[t] |= w;
                all_cols[t] = r == 0 ? w : all_cols[t] & w;
                pairs += __builtin_popcountll(w);
            }
            if (!any) {
                skip |= 1u << r;
                uniform.push_back(row0 + r);
            }
        }
        rbinfo[(size_t)rb * 2] = (int32_t)skip;
        rbinfo[(size_t)rb * 2 + 1] = row0;
        // tiles: 64 keys from the first key some row of the block may attend, then from the next such key, ...
        for (int c = 0; c < L;) {
            int t = c >> 6;
            uint64_t w = t < nw ? uni_cols[t] & (~0ull << (c & 63)) : 0;
            while (!w && ++t < nw) w = uni_cols[t];
            if (!w) break;
            const int key0 = t * 64 + __builtin_ctzll(w);
            const uint64_t all = key0 + KT <= L ? window(all_cols.data(), nw, key0) : 0ull;
            items.push_back(key0);
            if (all == ~0ull) {
                items.push_back(-1);
            } else {
                uint64_t m[RB];
                uint64_t hsh = 0x9e3779b97f4a7c15ull;
                for (int r = 0; r < RB; ++r) {
                    m[r] = r < rows ? window(bits.data() + (size_t)r * nw, nw, key0) : 0ull;
                    hsh = (hsh ^ m[r]) * 0xff51afd7ed558ccdull;
                    hsh ^= hsh >> 29;
                }
                int32_t found = -1;
                auto at = mask_of.find(hsh);
                if (at != mask_of.end())
                    for (int32_t i = at->second; i >= 0 && found < 0; i = next_same[i])
                        if (!std::memcmp(masks.data() + (size_t)i * RB, m, sizeof(m))) found = i;
                if (found < 0) {
                    found = (int32_t)(masks.size() / RB);
                    masks.insert(masks.end(), m, m + RB);
                    next_same.push_back(at != mask_of.end() ? at->second : -1);
                    mask_of[hsh] = found;
                }
                items.push_back(found);
            }
            c = key0 + KT;
        }
        rb_item_begin[rb + 1] = (int32_t)(items.size() / 2);
        for (int32_t t : tw) {  // leave the buffers zero for the next block
            for (int r = 0; r < rows; ++r) bits[(size_t)r * nw + t] = 0ull;
            uni_cols[t] = 0ull;
            all_cols[t] = 0ull;
            touched[t] = 0;
        }
    }

    const int n_items = (int)(items.size() / 2);
    int T = max_items_per_unit;
    if (T <= 0) {
        // Default: keep enough units in flight to cover the chip for one KV head group, but never make a
        // unit so short that the merge of its partial result costs more than walking the tiles.
        T = (n_items + 47) / 48;
        T = std::max(8, std::min(64, T));
    }

    struct Unit { int32_t rb, b, e, slot; };
    std::vector<Unit> units;
    std::vector<int32_t> split;
    int n_slots = 0;
    for (int rb = 0; rb < nrb; ++rb) {
        const int b = rb_item_begin[rb], e = rb_item_begin[rb + 1], n = e - b;
        if (n == 0) continue;
        if (n <= T) {
            units.push_back({rb, b, e, -1});
            continue;
        }
        const int nch = (n + T - 1) / T;
        const int per = (n + nch - 1) / nch;
        split.push_back(rb_row0[rb]);
        split.push_back(n_slots);
        int made = 0;
        for (int s = b; s < e; s += per, ++made) units.push_back({rb, s, std::min(e, s + per), n_slots + made});
        split.push_back(made);
        split.push_back(rb_rows(rb));
        n_slots += made;
    }
    std::stable_sort(units.begin(), units.end(), [](const Unit& a, const Unit& c) { return (a.e - a.b) > (c.e - c.b); });

    int64_t off = H_WORDS;
    const int64_t off_units = off;  off += (int64_t)units.size() * UNIT_WORDS;
    const int64_t off_items = off;  off += (int64_t)n_items * 2;
    if (off & 1) ++off;
    const int64_t off_masks = off;  off += (int64_t)masks.size() * 2;
    const int64_t off_rbinfo = off; off += (int64_t)nrb * 2;
    const int64_t off_split = off;  off += (int64_t)split.size();
    const int64_t off_uniform = off; off += (int64_t)uniform.size();
    off = align_up(off, 16);
    const int64_t off_lmasks = off; off += (int64_t)masks.size() * 2;
    const int64_t off_streams = off; off += n_streams + 1;
    off = align_up(off, 8);
    const int64_t off_steps = off;  off += (int64_t)n_items * STEP_WORDS;
    const int64_t total = align_up(off, 16);
    if (n_words) *n_words = total;
    if (total > 0x7fffffff) return fail(SDAG_ERR_ARG, "descriptor too large (%lld words)", (long long)total);
    if (!out || cap_words < total) return fail(SDAG_ERR_CAPACITY, "descriptor needs %lld words, buffer has %lld", (long long)total, (long long)cap_words);

    std::memset(out, 0, sizeof(int32_t) * (size_t)total);
    out[H_MAGIC] = DESC_MAGIC;
    out[H_VERSION] = DESC_VERSION;
    out[H_L] = L;
    out[H_NRB] = nrb;
    out[H_NKT] = nw;
    out[H_NUNITS] = (int32_t)units.size();
    out[H_NITEMS] = n_items;
    out[H_NMASKS] = (int32_t)(masks.size() / RB);
    out[H_NSPLIT] = (int32_t)(split.size() / 4);
    out[H_NUNIFORM] = (int32_t)uniform.size();
    out[H_OFF_UNITS] = (int32_t)off_units;
    out[H_OFF_ITEMS] = (int32_t)off_items;
    out[H_OFF_MASKS] = (int32_t)off_masks;
    out[H_OFF_RBINFO] = (int32_t)off_rbinfo;
    out[H_OFF_SPLIT] = (int32_t)off_split;
    out[H_OFF_UNIFORM] = (int32_t)off_uniform;
    out[H_TOTAL_WORDS] = (int32_t)total;
    out[H_NSLOTS] = n_slots;
    out[H_PAIRS_LO] = (int32_t)(uint32_t)(pairs & 0xffffffffll);
    out[H_PAIRS_HI] = (int32_t)(uint32_t)((uint64_t)pairs >> 32);
    out[H_NSTREAMS] = n_streams;
    out[H_NSTEPS] = n_items;
    out[H_OFF_LMASKS] = (int32_t)off_lmasks;
    out[H_OFF_STREAMS] = (int32_t)off_streams;
    out[H_OFF_STEPS] = (int32_t)off_steps;
    for (size_t u = 0; u < units.size(); ++u) {
        int32_t* p = out + off_units + u * UNIT_WORDS;
        p[0] = units[u].rb; p[1] = units[u].b; p[2] = units[u].e; p[3] = units[u].slot;
        p[4] = rbinfo[(size_t)units[u].rb * 2];
        p[5] = items[(size_t)units[u].b * 2];
        p[6] = items[(size_t)units[u].b * 2 + 1];
        p[7] = rb_row0[units[u].rb];
    }
    if (n_items) std::memcpy(out + off_items, items.data(), sizeof(int32_t) * items.size());
    if (off_masks > off_items + (int64_t)n_items * 2) out[off_masks - 1] = 0;
    if (!masks.empty()) std::memcpy(out + off_masks, masks.data(), sizeof(uint64_t) * masks.size());
    if (nrb) std::memcpy(out + off_rbinfo, rbinfo.data(), sizeof(int32_t) * rbinfo.size());
    if (!split.empty()) std::memcpy(out + off_split, split.data(), sizeof(int32_t) * split.size());
    if (!uniform.empty()) std::memcpy(out + off_uniform, uniform.data(), sizeof(int32_t) * uniform.size());
    for (size_t m = 0; m < masks.size() / RB; ++m)
        lane_mask_image(masks.data() + m * RB, reinterpret_cast<uint64_t*>(out + off_lmasks) + m * 32);
    std::vector<SchedUnit> su;
    su.reserve(units.size());
    for (const Unit& u : units)
        su.push_back({0, rb_row0[u.rb], u.b, u.e, u.slot, rbinfo[(size_t)u.rb * 2], L, items.data(), off_lmasks});
    build_schedule(su, n_streams, out + off_streams, out + off_steps);
    return SDAG_OK;
}

}  // namespace
}  // namespace sdag

using namespace sdag;

extern "C" int sdag_version(void) { return 1; }
extern "C" const char* sdag_last_error(void) { return err_buf(); }

extern "C" int sdag_desc_build(int L, int sys_user_len, int n_docs, const int32_t* spans, int qa_start,
                               const int32_t* nbr_off, const int32_t* nbr_idx, int max_items_per_unit,
                               int n_streams, int32_t* out, int64_t cap_words, int64_t* n_words) {
    if (L < 0 || n_docs < 0) return fail(SDAG_ERR_ARG, "negative L or n_docs");
    if (sys_user_len < 0 || sys_user_len > L) return fail(SDAG_ERR_ARG, "sys_user_len %d outside [0, %d]", sys_user_len, L);
    if (qa_start < 0) return fail(SDAG_ERR_ARG, "qa_start negative");
    if (n_docs > 0 && !spans) return fail(SDAG_ERR_ARG, "spans is NULL");
    for (int d = 0; d < n_docs; ++d) {
        const int s = spans[2 * d], e = spans[2 * d + 1];
        if (s < 0 || e < 0 || s > L || e > L) return fail(SDAG_ERR_ARG, "document %d span (%d, %d) outside [0, %d]", d, s, e, L);
    }
    if (nbr_off) {
        if (nbr_off[0] != 0) return fail(SDAG_ERR_ARG, "nbr_off[0] must be 0");
        for (int d = 0; d < n_docs; ++d)
            if (nbr_off[d + 1] < nbr_off[d]) return fail(SDAG_ERR_ARG, "nbr_off not monotone");
        if (n_docs > 0 && nbr_off[n_docs] > 0 && !nbr_idx) return fail(SDAG_ERR_ARG, "nbr_idx is NULL");
    }
    const int nkt = (L + SDAG_KV_TILE - 1) / SDAG_KV_TILE;
    auto fill_row = [&](int i, uint64_t* row, uint8_t* touched) {
        if (i < sys_user_len || i >= qa_start) set_range(row, touched, 0, i + 1);  // SDAG.py:107-108, 124-125
        for (int d = 0; d < n_docs; ++d) {
            const int s = spans[2 * d], e = spans[2 * d + 1];
            if (i < s || i >= e) continue;
            if (sys_user_len > 0) set_range(row, touched, 0, sys_user_len);  // SDAG.py:115
            set_range(row, touched, s, i + 1);                               // SDAG.py:116
            if (nbr_off) {
                for (int t = nbr_off[d]; t < nbr_off[d + 1]; ++t) {  // SDAG.py:118-122
                    const int nb = nbr_idx[t];
                    if (nb < 0 || nb >= n_docs) continue;
                    const int ns = spans[2 * nb], ne = spans[2 * nb + 1];
                    if (ne > ns) set_range(row, touched, ns, ne);
                }
            }
        }
    };
    // which of the row's conditions above hold: bit 0 the causal prefix, bit 1 + (d % 62) inside document d (documents
    // beyond 62 share bits: equal signatures of two rows then do not prove equal conditions, and are not taken to)
    auto signature = [&](int i) {
        uint64_t sg = (i < sys_user_len || i >= qa_start) ? 1ull : 0ull;
        for (int d = 0; d < n_docs; ++d)
            if (i >= spans[2 * d] && i < spans[2 * d + 1]) sg |= 2ull << (d % 62);
        return sg;
    };
    std::vector<int32_t> words;  // the words the block's rows have bits in
    auto fill = [&](int r0, int rows, uint64_t* bits, uint8_t* touched) {
        // The rows of a block lie in one segment of the prompt (the cuts below), so the same conditions hold for all of
        // them and row i + 1 allows what row i allows plus key i + 1: the first row is built from the ranges, every
        // further one is its predecessor's words and one more bit.  (Rows that do not share the first row's conditions
        // -- a caller's cuts would have to differ from those below -- are built from the ranges one by one.)
        fill_row(r0, bits, touched);
        if (rows == 1) return;
        const uint64_t sg = signature(r0);
        if (n_docs > 62 || signature(r0 + rows - 1) != sg) {
            for (int i = r0 + 1; i < r0 + rows; ++i) fill_row(i, bits + (size_t)(i - r0) * nkt, touched);
            return;
        }
        words.clear();
        for (int t = 0; t < nkt; ++t)
            if (touched[t]) words.push_back(t);
        for (int i = r0 + 1; i < r0 + rows; ++i) {
            uint64_t* row = bits + (size_t)(i - r0) * nkt;
            const uint64_t* prev = row - nkt;
            for (int32_t t : words) row[t] = prev[t];
            if (sg) {  // (no condition holds: the rows allow nothing)
                const int t = i >> 6;
                if (!touched[t]) {
                    touched[t] = 1;
                    words.push_back(t);
                }
                row[t] |= 1ull << (i & 63);
            }
        }
    };
    // row blocks start with the prompt's segments: system text, every document (and the gap behind it), the question
    std::vector<int> cuts = {sys_user_len, qa_start};
    for (int d = 0; d < n_docs; ++d) {
        cuts.push_back(spans[2 * d]);
        cuts.push_back(spans[2 * d + 1]);
    }
    return emit(L, cuts, fill, max_items_per_unit, n_streams, out, cap_words, n_words);
}

extern "C" int sdag_desc_from_dense(const uint8_t* mask, int L, int max_items_per_unit, int n_streams, int32_t* out,
                                    int64_t cap_words, int64_t* n_words) {
    if (L < 0 || (L > 0 && !mask)) return fail(SDAG_ERR_ARG, "bad dense mask");
    const int nkt = (L + SDAG_KV_TILE - 1) / SDAG_KV_TILE;
    auto fill = [&](int r0, int rows, uint64_t* bits, uint8_t* touched) {
        for (int i = r0; i < r0 + rows; ++i) {
            const uint8_t* src = mask + (size_t)i * L;
            uint64_t* row = bits + (size_t)(i - r0) * nkt;
            for (int j = 0; j < L; ++j)
                if (src[j]) {
                    row[j >> 6] |= 1ull << (j & 63);
                    touched[j >> 6] = 1;
                }
        }
    };
    return emit(L, {}, fill, max_items_per_unit, n_streams, out, cap_words, n_words);
}

extern "C" int sdag_desc_stats(const int32_t* d, int64_t* allowed_pairs, int32_t* n_uniform_rows,
                               int32_t* n_units, int32_t* n_items, int32_t* n_partial_tiles) {
    if (!desc_ok(d)) return fail(SDAG_ERR_DESC, "not a descriptor");
    if (allowed_pairs) *allowed_pairs = (int64_t)(((uint64_t)(uint32_t)d[H_PAIRS_HI] << 32) | (uint32_t)d[H_PAIRS_LO]);
    if (n_uniform_rows) *n_uniform_rows = d[H_NUNIFORM];
    if (n_units) *n_units = d[H_NUNITS];
    if (n_items) *n_items = d[H_NITEMS];
    if (n_partial_tiles) *n_partial_tiles = d[H_NMASKS];
    return SDAG_OK;
}

extern "C" int sdag_batch_build(const int32_t* const* descs, int B, int n_streams, int32_t* out, int64_t cap_words,
                                int64_t* n_words) {
    if (B <= 0 || !descs) return fail(SDAG_ERR_ARG, "bad batch");
    if (n_streams <= 0) n_streams = DEFAULT_STREAMS;
    if (n_streams > 65536) return fail(SDAG_ERR_ARG, "too many streams");
    for (int b = 0; b < B; ++b)
        if (!desc_ok(descs[b])) return fail(SDAG_ERR_DESC, "sequence %d: not a descriptor", b);
    std::vector<int32_t> merge, uniform, slot_base(B), desc_off(B);
    int64_t n_slots = 0, n_units = 0, n_steps = 0;
    int lmax = 0;
    for (int b = 0; b < B; ++b) {
        const int32_t* d = descs[b];
        lmax = std::max(lmax, d[H_L]);
        slot_base[b] = (int32_t)n_slots;
        n_slots += d[H_NSLOTS];
        n_units += d[H_NUNITS];
        n_steps += d[H_NITEMS];
        for (int e = 0; e < d[H_NSPLIT]; ++e) { merge.push_back(b); merge.push_back(e); }
        const int32_t* un = d + d[H_OFF_UNIFORM];
        for (int i = 0; i < d[H_NUNIFORM]; ++i) { uniform.push_back(b); uniform.push_back(un[i]); }
    }
    int64_t off = BT_WORDS;
    const int64_t off_desc = off;     off += B;
    const int64_t off_slot = off;     off += B;
    const int64_t off_merge = off;    off += (int64_t)merge.size();
    const int64_t off_uniform = off;  off += (int64_t)uniform.size();
    const int64_t off_streams = off;  off += n_streams + 1;
    off = align_up(off, 8);
    const int64_t off_steps = off;    off += n_steps * STEP_WORDS;
    // descriptors: one copy per distinct pointer, each at a multiple of 16 words (their lane masks are read in
    // 64-byte pieces)
    std::vector<const int32_t*> uniq;
    for (int b = 0; b < B; ++b) {
        size_t j = 0;
        while (j < uniq.size() && uniq[j] != descs[b]) ++j;
        if (j == uniq.size()) {
            uniq.push_back(descs[b]);
            off = align_up(off, 16);
            desc_off[b] = (int32_t)off;
            off += descs[b][H_OFF_STEPS];  // without its own step records (the last part): the batch has its own
        } else {
            for (int c = 0; c < b; ++c)
                if (descs[c] == descs[b]) { desc_off[b] = desc_off[c]; break; }
        }
    }
    // The lane masks the steps point at: ONE pool for the batch, every distinct mask once.  Prompts of like documents
    // share most of their masks (tiles follow the documents, so a document's staircase looks the same in every prompt):
    // the workgroups of a compute unit, busy with several sequences at a time, then pull a few KiB through the scalar
    // cache instead of every sequence's own copies.
    std::vector<std::vector<int32_t>> lmask_at((size_t)B);
    std::vector<const int32_t*> pool;  // source of every pooled mask (64 words)
    {
        std::unordered_map<uint64_t, std::vector<int32_t>> seen;  // hash -> pool indices
        off = align_up(off, 16);
        for (int b = 0; b < B; ++b) {
            const int32_t* d = descs[b];
            int first = 0;
            while (descs[first] != d) ++first;
            if (first < b) { lmask_at[b] = lmask_at[first]; continue; }
            const int32_t* lm = d + d[H_OFF_LMASKS];
            lmask_at[b].resize((size_t)d[H_NMASKS]);
            for (int m = 0; m < d[H_NMASKS]; ++m) {
                const int32_t* src = lm + (size_t)m * 64;
                uint64_t hsh = 0x9e3779b97f4a7c15ull;
                for (int i = 0; i < 32; ++i) {
                    uint64_t w;
                    std::memcpy(&w, src + 2 * i, 8);
                    hsh = (hsh ^ w) * 0xff51afd7ed558ccdull;
                    hsh ^= hsh >> 29;
                }
                int32_t found = -1;
                std::vector<int32_t>& cands = seen[hsh];
                for (int32_t c : cands)
                    if (!std::memcmp(pool[(size_t)c], src, 256)) { found = c; break; }
                if (found < 0) {
                    found = (int32_t)pool.size();
                    pool.push_back(src);
                    cands.push_back(found);
                }
                lmask_at[b][(size_t)m] = (int32_t)(off + (int64_t)found * 64);
            }
        }
    }
    const int64_t off_pool = off;
    off += (int64_t)pool.size() * 64;
    const int64_t total = align_up(off, 16);
    if (n_words) *n_words = total;
    if (total > 0x7fffffff || n_slots > 0x7fffffff) return fail(SDAG_ERR_ARG, "batch table too large");
    if (!out || cap_words < total) return fail(SDAG_ERR_CAPACITY, "batch table needs %lld words, buffer has %lld", (long long)total, (long long)cap_words);
    // Zero what nothing below writes: the head of the table up to the step records (every word of those is written by
    // build_schedule) and the few alignment words around the sequences' copies -- not the megabytes in between.
    std::memset(out, 0, sizeof(int32_t) * (size_t)off_steps);
    {
        int64_t end = off_steps + n_steps * STEP_WORDS;  // end of what has been accounted for so far
        for (const int32_t* u : uniq) {
            int b = 0;
            while (descs[b] != u) ++b;
            for (int64_t w = end; w < desc_off[b]; ++w) out[w] = 0;
            end = (int64_t)desc_off[b] + u[H_OFF_STEPS];
        }
        for (int64_t w = end; w < off_pool; ++w) out[w] = 0;
        for (int64_t w = off_pool + (int64_t)pool.size() * 64; w < total; ++w) out[w] = 0;
    }
    for (size_t i = 0; i < pool.size(); ++i) std::memcpy(out + off_pool + (int64_t)i * 64, pool[i], 256);
    out[BT_MAGIC] = BATCH_MAGIC;
    out[BT_VERSION] = BATCH_VERSION;
    out[BT_B] = B;
    out[BT_NPAIRS] = (int32_t)n_units;
    out[BT_NMERGE] = (int32_t)(merge.size() / 2);
    out[BT_NUNIFORM] = (int32_t)(uniform.size() / 2);
    out[BT_NSLOTS] = (int32_t)n_slots;
    out[BT_OFF_DESC] = (int32_t)off_desc;
    out[BT_OFF_SLOTBASE] = (int32_t)off_slot;
    out[BT_OFF_MERGE] = (int32_t)off_merge;
    out[BT_OFF_UNIFORM] = (int32_t)off_uniform;
    out[BT_TOTAL_WORDS] = (int32_t)total;
    out[BT_LMAX] = lmax;
    out[BT_NSTREAMS] = n_streams;
    out[BT_NSTEPS] = (int32_t)n_steps;
    out[BT_OFF_STREAMS] = (int32_t)off_streams;
    out[BT_OFF_STEPS] = (int32_t)off_steps;
    for (int b = 0; b < B; ++b) {
        out[off_desc + b] = desc_off[b];
        out[off_slot + b] = slot_base[b];
    }
    if (!merge.empty()) std::memcpy(out + off_merge, merge.data(), sizeof(int32_t) * merge.size());
    if (!uniform.empty()) std::memcpy(out + off_uniform, uniform.data(), sizeof(int32_t) * uniform.size());
    // (a sequence's copy keeps its header as it is: H_TOTAL_WORDS and H_OFF_STEPS still describe the full descriptor,
    // whose step records are not part of the copy and are never read through a batch table)
    for (int b = 0; b < B; ++b) std::memcpy(out + desc_off[b], descs[b], sizeof(int32_t) * (size_t)descs[b][H_OFF_STEPS]);
    // Sequence-major order (units inside a descriptor are already longest first): the workgroups that read one
    // sequence's K/V run together, so those rows stay in the XCD's L2 while they are needed.
    std::vector<SchedUnit> su;
    su.reserve((size_t)n_units);
    for (int b = 0; b < B; ++b) {
        const int32_t* d = descs[b];
        const int32_t* u = d + d[H_OFF_UNITS];
        for (int i = 0; i < d[H_NUNITS]; ++i, u += UNIT_WORDS)
            su.push_back({b, u[7], u[1], u[2], u[3] < 0 ? -1 : slot_base[b] + u[3], u[4], d[H_L], d + d[H_OFF_ITEMS],
                          (int64_t)desc_off[b] + d[H_OFF_LMASKS], lmask_at[b].data()});
    }
    // A batch with plenty of units per stream: the units of every sequence are cut in two by the keys they end at
    // (the first documents | the last documents and the rows that see everything), at equal cost, and each part goes
    // to one half of the streams.  Streams of one half then share their K/V rows with nobody outside it.  Why: the
    // two workgroups a compute unit holds do not advance at the same speed (the one dispatched first wins the issue
    // arbitration), blocks are dispatched in index order, so the first half of the streams runs ahead of the second by
    // several sequences; with every stream reading every document, the rows a lagging stream needs have long left the
    // L2 of its XCD.  Nothing depends on that dispatch order: without it the schedule is as balanced as the plain one.
    std::vector<uint8_t> half;
    if (n_streams >= 2 && n_streams % 2 == 0 && n_units >= 8 * (int64_t)n_streams) {
        half.assign(su.size(), 0);
        int64_t load2[2] = {0, 0};
        size_t first = 0;
        for (int b = 0; b < B; ++b) {
            const size_t n_b = (size_t)descs[b][H_NUNITS];
            std::vector<size_t> idx(n_b);
            for (size_t k = 0; k < n_b; ++k) idx[k] = first + k;
            auto last_key = [&](size_t u) { return su[u].items[2 * (su[u].item_end - 1)]; };
            std::stable_sort(idx.begin(), idx.end(), [&](size_t x, size_t y) { return last_key(x) < last_key(y); });
            int64_t total = 0;
            for (size_t u : idx) total += unit_cost(su[u]);
            // the part that ends at the low keys goes to the half with less work so far, up to an even overall split
            const int low = load2[0] <= load2[1] ? 0 : 1;
            const int64_t want = (total + load2[1 - low] - load2[low]) / 2;
            int64_t acc = 0;
            for (size_t u : idx) {
                const int h = acc + unit_cost(su[u]) / 2 <= want ? low : 1 - low;
                half[u] = (uint8_t)h;
                load2[h] += unit_cost(su[u]);
                acc += unit_cost(su[u]);
            }
            first += n_b;
        }
    }
    if (!half.empty()) {
        // ... and inside a sequence the units are dealt in the order of the keys they end at: the row blocks of one
        // document then run at the same time on different streams, and its K/V rows are fetched into the L2 once for
        // all of them (dealt longest first, they are spread over the time the sequence takes)
        std::vector<size_t> perm(su.size());
        for (size_t i = 0; i < perm.size(); ++i) perm[i] = i;
        std::stable_sort(perm.begin(), perm.end(), [&](size_t x, size_t y) {
            if (su[x].b != su[y].b) return su[x].b < su[y].b;
            // (long units - rows that see most of the prompt - first: dealt last, they would unbalance the streams)
            const bool lx = unit_is_long(su[x]), ly = unit_is_long(su[y]);
            if (lx != ly) return lx;
            return su[x].items[2 * (su[x].item_end - 1)] < su[y].items[2 * (su[y].item_end - 1)];
        });
        std::vector<SchedUnit> su2;
        std::vector<uint8_t> h2;
        su2.reserve(su.size());
        h2.reserve(su.size());
        for (size_t i : perm) {
            su2.push_back(su[i]);
            h2.push_back(half[i]);
        }
        su.swap(su2);
        half.swap(h2);
    }
    build_schedule(su, n_streams, out + off_streams, out + off_steps, half.empty() ? nullptr : &half);
    return SDAG_OK;
}
