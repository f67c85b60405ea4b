// Shared definitions of libsdag_hip.so: error reporting and the descriptor word layout.
#pragma once

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/sdag_hip.h"

namespace sdag {

// Thread-local message behind sdag_last_error().
char* err_buf();
int fail(int code, const char* fmt, ...);

#define SDAG_HIP_CHECK(expr)                                                                  \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            return ::sdag::fail(SDAG_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

// ---- descriptor layout (int32 words) -------------------------------------------------------------
// A descriptor describes one L x L "may attend" relation as tiles of at most 32 query rows x 64 keys.  The rows are
// cut into row blocks of up to 32 consecutive rows that start at the boundaries of the prompt's segments (system
// text, every document, the question), so a block never straddles two documents; the keys a block may attend are
// covered by 64-key tiles that start at the first allowed key of every run, so a document's tiles start with the
// document.  header | units | items | masks | rbinfo | split | uniform | lane masks | stream offsets | steps
constexpr int32_t DESC_MAGIC = 0x47414453;  // "SDAG"
constexpr int32_t DESC_VERSION = 4;
enum : int {
    H_MAGIC = 0,
    H_VERSION = 1,
    H_L = 2,
    H_NRB = 3,        // row blocks
    H_NKT = 4,        // ceil(L / 64)
    H_NUNITS = 5,     // work units (one workgroup each)
    H_NITEMS = 6,     // (key tile, mask) items over all units
    H_NMASKS = 7,     // partial-tile bit masks
    H_NSPLIT = 8,     // row blocks processed by more than one unit
    H_NUNIFORM = 9,   // rows that allow nothing (uniform attention over all keys)
    H_OFF_UNITS = 10,
    H_OFF_ITEMS = 11,
    H_OFF_MASKS = 12,  // even word offset: masks are read as 64-bit words
    H_OFF_RBINFO = 13,
    H_OFF_SPLIT = 14,
    H_OFF_UNIFORM = 15,
    H_TOTAL_WORDS = 16,
    H_NSLOTS = 17,  // partial-result slots (sum of units over split row blocks)
    H_PAIRS_LO = 18,
    H_PAIRS_HI = 19,
    H_NSTREAMS = 20,     // work streams (one persistent workgroup per stream and key/value head slot)
    H_NSTEPS = 21,       // steps over all streams (= items)
    H_OFF_LMASKS = 22,   // multiple of 16 words
    H_OFF_STREAMS = 23,  // n_streams + 1 step indices
    H_OFF_STEPS = 24,    // multiple of 8 words
    H_WORDS = 32
};
// unit   : 8 words {row block index, item_begin, item_end, slot (-1 = writes the output itself), skip bits of the
//          row block, first key and mask index of the first item, first row of the row block}
constexpr int UNIT_WORDS = 8;
// item   : 2 words {first key of the tile, mask_index (-1 = every valid row of the block may attend all 64 keys)}
// mask   : 32 rows x uint64, bit c of row r = row (row0 + r) may attend key (key0 + c)
// rbinfo : 2 words per row block {skip bits: rows that allow nothing or lie beyond the block's last row, first row};
//          a block ends where the next one starts (the last one at L)
// split  : 4 words {first row, first_slot, n_slots, rows in the block}
// uniform: 1 word per row that allows nothing: its index
// lane mask: the same 32 x 64 bits as a mask, regrouped for the attention kernel: 32 uint64, word 16 kb + i holds for
//          accumulator register i of key block kb (keys 32 kb .. 32 kb + 31 of the tile) one bit per lane:
//          bit (r + 32 h) = row r may attend key 32 kb + (i & 3) + 8 (i >> 2) + 4 h
// stream : the units are dealt to n_streams work streams (greedy, least loaded first); a stream is a run of steps
// step   : 8 words, one per (unit, item) in the order a workgroup walks them
constexpr int STEP_WORDS = 8;
enum : int {
    S_KEY0 = 0,   // first key of the tile
    S_MOFF = 1,   // word offset of the tile's lane mask from the start of the table, 0 = every pair allowed
    S_FLAGS = 2,  // bit 0: first step of its unit, bit 1: last step of its unit
    S_B = 3,      // sequence (0 in a single descriptor)
    S_L = 4,      // length of that sequence
    S_ROW0 = 5,   // first row of the row block
    S_SLOT = 6,   // partial-result slot (global inside a batch table) or -1
    S_UBITS = 7   // skip bits of the row block: rows whose output this step's unit must not write
};
constexpr int DEFAULT_STREAMS = 64;

// ---- batch table (int32 words): several sequences, each with its own descriptor, in one launch ------------
// header | desc_off[B] | slot_base[B] | merge {sequence, split index} | uniform {sequence, row} | stream offsets |
// steps (as in a descriptor, with S_B the sequence, S_SLOT global and S_MOFF relative to the table) | the
// descriptors themselves (each at a multiple of 16 words, copied up to their own step records, which a batch
// table never uses: their headers still describe the full descriptor).  Units are dealt to the streams in sequence-major
// order, longest first inside a sequence, so the workgroups running at any time read the K/V of few sequences.
constexpr int32_t BATCH_MAGIC = 0x54424453;  // "SDBT"
enum : int {
    BT_MAGIC = 0,
    BT_VERSION = 1,
    BT_B = 2,
    BT_NPAIRS = 3,
    BT_NMERGE = 4,
    BT_NUNIFORM = 5,
    BT_NSLOTS = 6,
    BT_OFF_DESC = 7,
    BT_OFF_SLOTBASE = 8,
    BT_RESERVED9 = 9,
    BT_OFF_MERGE = 10,
    BT_OFF_UNIFORM = 11,
    BT_TOTAL_WORDS = 12,
    BT_LMAX = 13,
    BT_NSTREAMS = 14,
    BT_NSTEPS = 15,
    BT_OFF_STREAMS = 16,
    BT_OFF_STEPS = 17,
    BT_WORDS = 32
};
constexpr int32_t BATCH_VERSION = 3;
inline bool batch_ok(const int32_t* d) { return d && d[BT_MAGIC] == BATCH_MAGIC && d[BT_VERSION] == BATCH_VERSION; }

inline bool desc_ok(const int32_t* d) { return d && d[H_MAGIC] == DESC_MAGIC && d[H_VERSION] == DESC_VERSION; }

}  // namespace sdag
