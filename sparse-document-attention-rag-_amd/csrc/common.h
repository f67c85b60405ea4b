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
// A descriptor describes one L x L "may attend" relation at a granularity of 32-row blocks and 64-key
// tiles.  header | units | items | masks | rbinfo | split | uniform
constexpr int32_t DESC_MAGIC = 0x47414453;  // "SDAG"
constexpr int32_t DESC_VERSION = 2;
enum : int {
    H_MAGIC = 0,
    H_VERSION = 1,
    H_L = 2,
    H_NRB = 3,        // row blocks
    H_NKT = 4,        // key tiles
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
    H_WORDS = 32
};
// unit   : 8 words {row_block, item_begin, item_end, slot (-1 = writes the output itself), uniform_rows bits of the
//          row block, key tile and mask index of the first item, 0}: all a workgroup needs to start loading
constexpr int UNIT_WORDS = 8;
// item   : 2 words {key_tile, mask_index (-1 = every pair of the tile allowed)}
// mask   : 32 rows x uint64, bit c of row r = row (32*rb + r) may attend key (64*kt + c)
// rbinfo : 2 words per row block {uniform_rows bits, reserved}
// split  : 4 words {row_block, first_slot, n_slots, 0}
// uniform: 1 word per row: its index

// ---- batch table (int32 words): several sequences, each with its own descriptor, in one launch ------------
// header | desc_off[B] | slot_base[B] | pairs (16 words each, sequence-major, longest units first inside: {sequence,
// row_block, item_begin, item_end, global slot or -1, uniform_rows bits, L, first key tile, first mask index, word
// offset of the sequence's items, word offset of its masks, 0...}) | merge {sequence, split index} |
// uniform {sequence, row} | the descriptors themselves (each at an even word offset)
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
    BT_OFF_PAIRS = 9,
    BT_OFF_MERGE = 10,
    BT_OFF_UNIFORM = 11,
    BT_TOTAL_WORDS = 12,
    BT_LMAX = 13,
    BT_WORDS = 16
};
constexpr int PAIR_WORDS = 16;
inline bool batch_ok(const int32_t* d) { return d && d[BT_MAGIC] == BATCH_MAGIC && d[BT_VERSION] == 1; }

inline bool desc_ok(const int32_t* d) { return d && d[H_MAGIC] == DESC_MAGIC && d[H_VERSION] == DESC_VERSION; }

}  // namespace sdag
