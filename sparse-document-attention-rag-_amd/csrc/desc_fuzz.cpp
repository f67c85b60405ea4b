// Sanitizer driver of the host-side descriptor builder (built and run by `make asan`: g++ with AddressSanitizer and
// UndefinedBehaviorSanitizer, no GPU, no Python -- the sanitizer runtime cannot be preloaded into the test interpreter).
//
// Random prompts (documents of 0-150 tokens, gaps, overlapping and empty spans, neighbour lists with out-of-range ids,
// random unit length limits and stream counts) go through sdag_desc_build into a buffer of EXACTLY the size the builder
// asked for, so any write past it is caught; the descriptor is then decoded from its documented word layout and must
// equal the mask the reference's rule gives (src/pipeline/sparse_attention_RAG/SDAG.py:104-127, restated below),
// every item must occur in exactly one step of one stream, and the lane-mask images must be the regrouped masks.
// Dense-mask descriptors (sdag_desc_from_dense) and batch tables (sdag_batch_build) are checked the same way.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <random>
#include <vector>

#include "common.h"

using namespace sdag;

#define CHECK(cond, ...)                                   \
    do {                                                   \
        if (!(cond)) {                                     \
            std::fprintf(stderr, "FAILED %s:%d: ", __FILE__, __LINE__); \
            std::fprintf(stderr, __VA_ARGS__);             \
            std::fprintf(stderr, "\n");                    \
            std::exit(1);                                  \
        }                                                  \
    } while (0)

struct Prompt {
    int L, sys_len, qa;
    std::vector<int32_t> spans;                 // 2 per document
    bool use_nbr;
    std::vector<int32_t> nbr_off, nbr_idx;
};

// The reference's rule, row by row.
static std::vector<uint8_t> reference_mask(const Prompt& p) {
    const int L = p.L, nd = (int)p.spans.size() / 2;
    std::vector<uint8_t> m((size_t)L * L, 0);
    for (int i = 0; i < L; ++i) {
        uint8_t* row = m.data() + (size_t)i * L;
        if (i < p.sys_len || i >= p.qa) std::fill(row, row + i + 1, 1);
        for (int d = 0; d < nd; ++d) {
            const int s = p.spans[2 * d], e = p.spans[2 * d + 1];
            if (i < s || i >= e) continue;
            std::fill(row, row + p.sys_len, 1);
            std::fill(row + s, row + i + 1, 1);
            if (p.use_nbr)
                for (int t = p.nbr_off[d]; t < p.nbr_off[d + 1]; ++t) {
                    const int nb = p.nbr_idx[t];
                    if (nb < 0 || nb >= nd) continue;
                    const int ns = p.spans[2 * nb], ne = p.spans[2 * nb + 1];
                    if (ne > ns) std::fill(row + ns, row + ne, 1);
                }
        }
    }
    return m;
}

// Decode a descriptor into a dense mask and check its internal consistency.
static std::vector<uint8_t> expand(const int32_t* w, int64_t n_words) {
    CHECK(desc_ok(w), "bad magic");
    CHECK(w[H_TOTAL_WORDS] == n_words, "total words %d != %lld", w[H_TOTAL_WORDS], (long long)n_words);
    const int L = w[H_L], nrb = w[H_NRB], n_items = w[H_NITEMS], n_units = w[H_NUNITS];
    std::vector<uint8_t> m((size_t)L * L, 0);
    const int32_t* units = w + w[H_OFF_UNITS];
    const int32_t* items = w + w[H_OFF_ITEMS];
    const uint64_t* masks = reinterpret_cast<const uint64_t*>(w + w[H_OFF_MASKS]);
    const uint64_t* lmasks = reinterpret_cast<const uint64_t*>(w + w[H_OFF_LMASKS]);
    const int32_t* rbinfo = w + w[H_OFF_RBINFO];
    CHECK((w[H_OFF_MASKS] & 1) == 0 && (w[H_OFF_LMASKS] & 15) == 0 && (w[H_OFF_STEPS] & 7) == 0, "alignment");
    std::vector<int> seen((size_t)n_items, 0);
    for (int u = 0; u < n_units; ++u) {
        const int32_t* un = units + (size_t)u * UNIT_WORDS;
        const int rb = un[0], b = un[1], e = un[2];
        CHECK(rb >= 0 && rb < nrb && b >= 0 && e <= n_items && b < e, "unit %d out of range", u);
        const int row0 = rbinfo[2 * rb + 1];
        const int rows = (rb + 1 < nrb ? rbinfo[2 * (rb + 1) + 1] : L) - row0;
        CHECK(rows > 0 && rows <= 32 && un[7] == row0, "row block %d", rb);
        for (int it = b; it < e; ++it) {
            CHECK(!seen[it]++, "item %d in two units", it);
            const int key0 = items[2 * it], mi = items[2 * it + 1];
            CHECK(key0 >= 0 && key0 < L, "key0 %d", key0);
            for (int r = 0; r < rows; ++r)
                for (int c = 0; c < 64 && key0 + c < L; ++c) {
                    const bool bit = mi < 0 ? true : ((masks[(size_t)mi * 32 + r] >> c) & 1ull) != 0;
                    if (bit) m[(size_t)(row0 + r) * L + key0 + c] = 1;
                }
            if (mi >= 0) {
                CHECK(mi < w[H_NMASKS], "mask index");
                for (int kb = 0; kb < 2; ++kb)
                    for (int i = 0; i < 16; ++i)
                        for (int h = 0; h < 2; ++h)
                            for (int r = 0; r < 32; ++r) {
                                const int key = 32 * kb + (i & 3) + 8 * (i >> 2) + 4 * h;
                                const uint64_t want = (masks[(size_t)mi * 32 + r] >> key) & 1ull;
                                const uint64_t got = (lmasks[(size_t)mi * 32 + 16 * kb + i] >> (r + 32 * h)) & 1ull;
                                CHECK(want == got, "lane mask %d", mi);
                            }
            } else {
                CHECK(key0 + 64 <= L, "a full tile reaches past L");
            }
        }
    }
    for (int it = 0; it < n_items; ++it) CHECK(seen[it] == 1, "item %d in no unit", it);
    // streams: every (unit, item) exactly once, unit steps consecutive, first / last flags right
    const int ns = w[H_NSTREAMS];
    const int32_t* so = w + w[H_OFF_STREAMS];
    const int32_t* steps = w + w[H_OFF_STEPS];
    CHECK(so[0] == 0 && so[ns] == n_items && w[H_NSTEPS] == n_items, "stream offsets");
    for (int s = 0; s < ns; ++s) CHECK(so[s] <= so[s + 1], "stream offsets not monotone");
    int64_t keysum = 0, keysum_items = 0;
    for (int i = 0; i < n_items; ++i) {
        const int32_t* st = steps + (size_t)i * STEP_WORDS;
        keysum += st[S_KEY0] + 7 * st[S_ROW0];
        CHECK(st[S_L] == L && st[S_B] == 0, "step %d", i);
        CHECK(st[S_MOFF] == 0 || (st[S_MOFF] >= w[H_OFF_LMASKS] && st[S_MOFF] + 64 <= w[H_OFF_STREAMS] && st[S_MOFF] % 16 == 0), "step mask offset");
    }
    for (int u = 0; u < n_units; ++u) {
        const int32_t* un = units + (size_t)u * UNIT_WORDS;
        for (int it = un[1]; it < un[2]; ++it) keysum_items += items[2 * it] + 7 * un[7];
    }
    CHECK(keysum == keysum_items, "steps do not cover the items");
    return m;
}

int main(int argc, char** argv) {
    const int n_cases = argc > 1 ? std::atoi(argv[1]) : 400;
    std::mt19937 rng(12345);
    auto ri = [&](int lo, int hi) { return lo + (int)(rng() % (unsigned)(hi - lo + 1)); };
    std::vector<std::unique_ptr<int32_t[]>> kept;   // descriptors for the batch tables
    std::vector<int64_t> kept_words;
    int64_t total_pairs = 0;
    for (int c = 0; c < n_cases; ++c) {
        Prompt p;
        const int nd = ri(0, 8);
        p.sys_len = ri(0, 70);
        int pos = p.sys_len;
        for (int d = 0; d < nd; ++d) {
            const int len = ri(0, 6) == 0 ? 0 : ri(1, 150);
            p.spans.push_back(pos);
            p.spans.push_back(pos + len);
            pos += len + ri(0, 3);
        }
        p.qa = pos;
        p.L = pos + ri(0, 40);
        if (p.L == 0) p.L = 1, p.qa = 0;
        if (nd >= 2 && ri(0, 4) == 0) {  // overlapping spans: the builder accepts them, later writes only add
            p.spans[2] = p.spans[0] + (p.spans[1] - p.spans[0]) / 2;
            if (p.spans[3] < p.spans[2]) p.spans[3] = p.spans[2];
        }
        p.use_nbr = ri(0, 2) != 0;
        p.nbr_off.push_back(0);
        for (int d = 0; d < nd; ++d) {
            const int k = ri(0, 3);
            for (int t = 0; t < k; ++t) p.nbr_idx.push_back(ri(-2, nd + 1));
            p.nbr_off.push_back((int32_t)p.nbr_idx.size());
        }
        const int T = ri(0, 5), ns = ri(0, 4) == 0 ? ri(1, 200) : 0;
        int64_t need = 0;
        int rc = sdag_desc_build(p.L, p.sys_len, nd, nd ? p.spans.data() : nullptr, p.qa, p.use_nbr ? p.nbr_off.data() : nullptr,
                                 p.use_nbr && !p.nbr_idx.empty() ? p.nbr_idx.data() : nullptr, T, ns, nullptr, 0, &need);
        CHECK(rc == SDAG_ERR_CAPACITY && need > 0, "size query: rc %d (%s)", rc, sdag_last_error());
        std::unique_ptr<int32_t[]> buf(new int32_t[(size_t)need]);
        if (need > 1) {  // one word short: refused, nothing written past the end
            rc = sdag_desc_build(p.L, p.sys_len, nd, nd ? p.spans.data() : nullptr, p.qa, p.use_nbr ? p.nbr_off.data() : nullptr,
                                 p.use_nbr && !p.nbr_idx.empty() ? p.nbr_idx.data() : nullptr, T, ns, buf.get(), need - 1, nullptr);
            CHECK(rc == SDAG_ERR_CAPACITY, "short buffer accepted");
        }
        int64_t got = 0;
        rc = sdag_desc_build(p.L, p.sys_len, nd, nd ? p.spans.data() : nullptr, p.qa, p.use_nbr ? p.nbr_off.data() : nullptr,
                             p.use_nbr && !p.nbr_idx.empty() ? p.nbr_idx.data() : nullptr, T, ns, buf.get(), need, &got);
        CHECK(rc == SDAG_OK && got == need, "build: rc %d (%s)", rc, sdag_last_error());
        const std::vector<uint8_t> ref = reference_mask(p);
        const std::vector<uint8_t> exp = expand(buf.get(), need);
        CHECK(ref == exp, "case %d: expanded descriptor differs from the reference rule (L %d, %d docs)", c, p.L, nd);
        int64_t pairs = 0;
        for (uint8_t x : ref) pairs += x;
        int64_t dp = 0;
        CHECK(sdag_desc_stats(buf.get(), &dp, nullptr, nullptr, nullptr, nullptr) == SDAG_OK && dp == pairs, "allowed pairs");
        total_pairs += pairs;
        // the same relation through the dense entry point
        if (c % 4 == 0) {
            int64_t need2 = 0;
            sdag_desc_from_dense(ref.data(), p.L, T, ns, nullptr, 0, &need2);
            std::unique_ptr<int32_t[]> b2(new int32_t[(size_t)need2]);
            rc = sdag_desc_from_dense(ref.data(), p.L, T, ns, b2.get(), need2, nullptr);
            CHECK(rc == SDAG_OK, "from_dense: %s", sdag_last_error());
            CHECK(expand(b2.get(), need2) == ref, "case %d: dense descriptor differs", c);
        }
        if (c % 7 == 0 && kept.size() < 40) {
            kept.push_back(std::move(buf));
            kept_words.push_back(need);
        }
    }
    // batch tables over random subsets (with repeats) of the kept descriptors
    for (int c = 0; c < 40 && !kept.empty(); ++c) {
        const int B = ri(1, 12);
        std::vector<const int32_t*> ptrs;
        int64_t items = 0;
        for (int b = 0; b < B; ++b) {
            ptrs.push_back(kept[(size_t)ri(0, (int)kept.size() - 1)].get());
            items += ptrs.back()[H_NITEMS];
        }
        const int ns = ri(0, 3) == 0 ? ri(1, 150) : 0;
        int64_t need = 0;
        int rc = sdag_batch_build(ptrs.data(), B, ns, nullptr, 0, &need);
        CHECK(rc == SDAG_ERR_CAPACITY && need > 0, "batch size query");
        std::unique_ptr<int32_t[]> bt(new int32_t[(size_t)need]);
        rc = sdag_batch_build(ptrs.data(), B, ns, bt.get(), need, nullptr);
        CHECK(rc == SDAG_OK && batch_ok(bt.get()), "batch build: %s", sdag_last_error());
        const int32_t* t = bt.get();
        CHECK(t[BT_TOTAL_WORDS] == need && t[BT_B] == B && t[BT_NSTEPS] == items, "batch header");
        const int32_t* so = t + t[BT_OFF_STREAMS];
        CHECK(so[0] == 0 && so[t[BT_NSTREAMS]] == items, "batch stream offsets");
        const int32_t* steps = t + t[BT_OFF_STEPS];
        std::vector<int64_t> per_seq((size_t)B, 0);
        for (int64_t i = 0; i < items; ++i) {
            const int32_t* st = steps + i * STEP_WORDS;
            const int b = st[S_B];
            CHECK(b >= 0 && b < B, "step sequence");
            const int32_t* d = t + t[t[BT_OFF_DESC] + b];
            CHECK(desc_ok(d) && st[S_L] == d[H_L], "step descriptor");
            CHECK(std::memcmp(d, ptrs[(size_t)b], sizeof(int32_t) * (size_t)d[H_OFF_STEPS]) == 0, "descriptor copy");
            if (st[S_MOFF]) {  // inside the table, at a 64-byte boundary, and one of the sequence's lane masks word for word
                CHECK(st[S_MOFF] > 0 && st[S_MOFF] % 16 == 0 && st[S_MOFF] + 64 <= need, "batch mask offset");
                const int32_t* lm = d + d[H_OFF_LMASKS];
                bool is_one = false;
                for (int m = 0; m < d[H_NMASKS] && !is_one; ++m) is_one = std::memcmp(t + st[S_MOFF], lm + (size_t)m * 64, 256) == 0;
                CHECK(is_one, "batch mask contents");
            }
            ++per_seq[(size_t)b];
        }
        for (int b = 0; b < B; ++b) CHECK(per_seq[(size_t)b] == ptrs[(size_t)b][H_NITEMS], "steps of sequence %d", b);
    }
    // argument errors must be refused, not crash
    {
        int32_t spans[2] = {5, 900};
        int64_t need = 0;
        CHECK(sdag_desc_build(100, 3, 1, spans, 50, nullptr, nullptr, 0, 0, nullptr, 0, &need) == SDAG_ERR_ARG, "span past L accepted");
        CHECK(sdag_desc_build(-1, 0, 0, nullptr, 0, nullptr, nullptr, 0, 0, nullptr, 0, &need) == SDAG_ERR_ARG, "negative L accepted");
        CHECK(sdag_desc_build(10, 20, 0, nullptr, 0, nullptr, nullptr, 0, 0, nullptr, 0, &need) == SDAG_ERR_ARG, "sys_len past L accepted");
        CHECK(sdag_batch_build(nullptr, 2, 0, nullptr, 0, &need) == SDAG_ERR_ARG, "NULL batch accepted");
    }
    std::printf("desc_fuzz ok: %d prompts, %lld allowed pairs checked\n", n_cases, (long long)total_pairs);
    return 0;
}
