// pmx_device.h -- structures shared by the HIP kernels and the C-ABI host code (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PMX_BLOCK 256            // expansion kernel: 4 wavefronts of 64
#ifndef PMX_RULE_BLOCK
#define PMX_RULE_BLOCK 64        // rule kernels: one wavefront per block, so that N/64 blocks spread over all CUs
#endif
#define PMX_SCARED_TIME 40       // capture.py:75
#define PMX_MIN_FOOD 2           // capture.py:70

// ---------------------------------------------------------------------------------------------
// Device state layout: structure-of-arrays of 32-bit words, word-major: state[word * N + env].
// One lane owns one env in the rule kernel, so every access below is a fully coalesced dword stream.
//   [0, H)        food rows, bit x of row y
//   H + i         agent i word A: x | y << 8 | dir << 16 | isPacman << 24
//   H + 4 + i     agent i word B: scaredTimer | numCarrying << 8 (12 bits) | numReturned << 20 (12 bits)
//   H + 8, H + 9  capsule list: four 16-bit slots (x | y << 8), 0xFFFF = empty
//   H + 10        score (int32)
//   H + 11        steps (int32)
//   H + 12        ticks since creation (never reset): counter of the PMX_ACTION_RANDOM_LEGAL generator
//   H + 13 ...    accumulators of an open tick, used only by pmx_step_agent:
//                 red reward (2 words, f64), blue reward (2 words), red score change, blue score change, total
// A "snapshot" is the first H + 10 words of this layout; the observation encoder reads snapshots.
// ---------------------------------------------------------------------------------------------
#define PMX_W_AGENT_A(H, i) ((H) + (i))
#define PMX_W_AGENT_B(H, i) ((H) + 4 + (i))
#define PMX_W_CAPS(H, j) ((H) + 8 + (j))
#define PMX_W_SCORE(H) ((H) + 10)
#define PMX_W_STEPS(H) ((H) + 11)
#define PMX_W_TICKS(H) ((H) + 12)
#define PMX_W_ACC(H) ((H) + 13)
#define PMX_SNAP_WORDS(H) ((H) + 10)
#define PMX_STATE_WORDS(H) ((H) + 20)

struct PmxLayoutDev {
    int32_t W, H, half;          // half = int(W / 2): the food split (capture.py:333)
    uint32_t lo_mask, hi_mask;   // columns x < half / x >= half
    uint32_t walls[32];
    uint32_t food0[32];
    uint32_t capw0[2];
    int32_t startx[4], starty[4];
    int32_t total_food;
    int32_t n_dump;              // entries of the dump-order table
    uint32_t wall_stream[32];    // plane 0 of the observation as a packed bit stream: bit (y*W + x) = wall
    int32_t n_cells;             // open cells (rows of this layout's maze-distance matrix)
    uint32_t dist_off;           // byte offset of that matrix in the handle's distance buffer
};

struct PmxTickParams {
    uint32_t *state;             // [PMX_STATE_WORDS][N]
    uint32_t *snap;              // [3][PMX_SNAP_WORDS][N] states after sub-steps 0,1,2
    const PmxLayoutDev *lay;     // [n_layouts]
    const int32_t *layout_idx;   // [N] layout of each env, or NULL when the handle has one layout
    const int8_t *dump;          // [n_dump][2] BFS visit order of dumpFoodFromDeath
    const int8_t *actions;
    int32_t N, length, legal_reward, defence_reward, auto_reset;
    double *reward;
    uint8_t *done;
    uint8_t *legal;
    int32_t *score_change;
    int32_t *score;
    const uint8_t *dist;         // maze-distance matrices of all layouts (in-kernel baselineTeam bots), or NULL
    const int16_t *cell_index;   // [n_layouts][32*32] cell -> row of the distance matrix
    uint32_t *agent_out;         // [N][4] x | y<<8 | carry<<16 right after the agent's own sub-step
    uint32_t seed;
    const uint8_t *reset_mask;   // reset kernel only: NULL = every env
    int32_t no_reset;            // reset kernel only: 1 = touch no env (legal masks of the current state only)
    // host-side copies of what every layout of a handle shares, so that the kernel has them with its arguments instead
    // of behind a dependent load from the layout record
    int32_t lay_W, lay_H, lay_half, lay_n_dump;
    uint32_t lo_mask, hi_mask;
    int32_t *layout_idx_rw;      // the same array as layout_idx when envs move to a new layout at every reset (redraw), else NULL
    int32_t n_layouts;
};

struct PmxExpandParams {
    const uint32_t *snap[4];     // per AGENT: base of the snapshot its planes are encoded from
    const PmxLayoutDev *lay;     // [n_layouts]
    const int32_t *layout_idx;   // [N] or NULL
    void *obs;                   // [N][n_emit][8][H][W]
    int32_t N, n_emit;
    int32_t emit[4];             // agent index of each emitted slot
    int32_t single_agent;        // >= 0: obs is [N][8][H][W] for that agent only (pmx_step_agent)
    int32_t lay_H, lay_W;        // host-side copies of the common layout dimensions (launch sizing)
    int32_t reverse;             // 1: walk the blocks from the highest address down (alternate ticks, see pmx_launch_expand)
};

// pmx_emit_team_obs: the two observations of one team, canonicalised for a red team, + the merged critic input
struct PmxEmitParams {
    const uint32_t *snap[4];     // per agent: the snapshot its planes are encoded from (as in PmxExpandParams)
    const PmxLayoutDev *lay;
    const int32_t *layout_idx;
    void *team_obs;              // [N][2][8][H][W]
    void *merged;                // [N][8][H][W] or NULL
    int32_t N, red;              // red: agents (0, 2), x-flipped with planes 2<->3 and 6<->7 swapped; else agents (1, 3) as they are
    int32_t lay_H, lay_W;
};

// Host-side launch tuning of the expansion kernel: -1 = the built-in choice.  Filled once per handle at pmx_create (from the
// PMX_EXPAND_* environment variables, for experiments) and changed through pmx_set_tuning; never read at launch time.
struct PmxExpandTuning {
    int32_t alt = -1;            // alternate the sweep direction from tick to tick
    int32_t nt = -1;             // non-temporal stores
    int32_t lds_pad = -1;        // dynamic-LDS reservation per block (occupancy cap), bytes
    int32_t lut = -1;            // LDS look-up-table expansion
    int32_t per_env = -1;        // one wave per env (pmx_expand4_kernel) instead of one per (env, agent)
};
