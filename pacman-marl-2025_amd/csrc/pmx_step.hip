// pmx_step.hip -- the batched Capture-the-Flag tick for gfx950 (MI355X).
//
// Two kernels per tick:
//   pmx_rule_kernel    one LANE per env: the four agent sub-steps of gymPacMan.step (gymPacMan.py:143-193) with the
//                      rules of capture.py:448-728, the shaped reward of gymPacMan.py:231-259 and the termination
//                      test of gymPacMan.py:261-270.  Food rows live in LDS (one column per lane, conflict free),
//                      agents in registers.  It leaves the state after each sub-step as a coalesced SoA snapshot.
//   pmx_expand_kernel  one WAVEFRONT per (env, agent): turns a snapshot into the 8 observation planes of
//                      gymPacMan.get_Observation (gymPacMan.py:195-229) with 16-byte-per-lane stores (ordinary while the
//                      planes can live in the Infinity Cache, walking the blocks in alternating directions from tick to
//                      tick; streaming with capped occupancy beyond).  This kernel moves >95 % of the bytes of the tick
//                      and is the HBM-roofline kernel.
// The split keeps the divergent integer rule logic at 64 envs per wave while the byte-heavy expansion gets
// N*4 wavefronts of perfectly coalesced stores regardless of N.
#include <cstdlib>

#include <hip/hip_ext.h>

#include "pmx_device.h"

#define PMX_MAX_H_LDS 32   // per-lane wall columns (multi-layout handles) start after room for 32 food rows

namespace {

struct Env {
    uint32_t xy[4];        // x | y << 8: one compare tests "same cell" (capture.py:682,708 collisions at tolerance 0.7)
    int dir[4], pac[4], scared[4], carry[4], ret[4];
    uint32_t capw[2];
    int score, steps;
    uint32_t ticks;        // never reset
    uint32_t self_after[4];  // x | y<<8 | carry<<16 of agent i right after its own sub-step
};

struct Acc {
    double red_r, blue_r;
    int red_sc, blue_sc, sc_total;
    int n_red, n_blue;     // pellets on the red / blue half (capture.py:332-342), kept up to date by the hot kernel only (HB > 0)
};

struct Ctx {
    const uint32_t *wl;   // LDS: wall rows, row y at wl[y * wls] (wls = 1: one shared layout; PMX_RULE_BLOCK: per-lane layouts)
    int wls;
    uint32_t *fd;         // LDS: this lane's food column, row y at fd[y * PMX_RULE_BLOCK]
    const PmxLayoutDev *L;
    const int8_t *dump;
    int W, H, half, n_dump;
    uint32_t lo_mask, hi_mask;   // halfGrid column masks of this layout (loaded once, up front)
    int legal_reward, defence_reward;
    uint32_t rng_key;     // seed ^ env * 0x9E3779B1 (the per-env part of the random-legal key)
    uint32_t start_xy[4]; // start cells, packed like Env::xy
    uint32_t *fd2;        // LDS: scratch food column for the reflex bots' what-if successors
    uint32_t *fd3;        // LDS: scratch column of dump_food (rows of cells that cannot take a pellet)
    const uint8_t *dist;  // this env's layout's maze-distance matrix (or NULL)
    const int16_t *cidx;  // its cell -> matrix row map
    int n_cells;
    uint32_t *rows0;      // LDS: row 0 of the whole block's food rows ([row][lane]; fd = rows0 + lane)
    uint32_t *stg;        // LDS: [16][PMX_RULE_BLOCK] staging words of the wide snapshot / state stores
};

__device__ __forceinline__ uint32_t pack_a(const Env &e, int i)
{
    return e.xy[i] | ((uint32_t)e.dir[i] << 16) | ((uint32_t)e.pac[i] << 24);
}
__device__ __forceinline__ uint32_t pack_b(const Env &e, int i)
{
    return (uint32_t)e.scared[i] | ((uint32_t)e.carry[i] << 8) | ((uint32_t)e.ret[i] << 20);
}
__device__ __forceinline__ void unpack_a(Env &e, int i, uint32_t w)
{
    e.xy[i] = w & 0xFFFFu; e.dir[i] = (w >> 16) & 0xFF; e.pac[i] = (w >> 24) & 1;
}
__device__ __forceinline__ void unpack_b(Env &e, int i, uint32_t w)
{
    e.scared[i] = w & 0xFF; e.carry[i] = (w >> 8) & 0xFFF; e.ret[i] = (w >> 20) & 0xFFF;
}

// capture.py:453-461 + game.py:335-350: bit a = action a legal (0 N, 1 E, 2 S, 3 W, 4 Stop)
__device__ __forceinline__ int legal_mask(const uint32_t *wl, int wls, int x, int y)
{
    uint32_t r0 = ~wl[y * wls], rn = ~wl[(y + 1) * wls], rs = ~wl[(y - 1) * wls];
    return (int)(((rn >> x) & 1u) | (((r0 >> (x + 1)) & 1u) << 1) | (((rs >> x) & 1u) << 2) |
                 (((r0 >> (x - 1)) & 1u) << 3) | (((r0 >> x) & 1u) << 4));
}

__device__ __forceinline__ int cap_find(const Env &e, uint32_t key)
{
    if ((e.capw[0] & e.capw[1]) == 0xFFFFFFFFu) return -1;      // no capsule left (the common case)
    if ((e.capw[0] & 0xFFFFu) == key) return 0;
    if ((e.capw[0] >> 16) == key) return 1;
    if ((e.capw[1] & 0xFFFFu) == key) return 2;
    if ((e.capw[1] >> 16) == key) return 3;
    return -1;
}
__device__ __forceinline__ void cap_remove(Env &e, int slot)
{
    uint32_t m = 0xFFFFu << (16 * (slot & 1));
    if (slot < 2) e.capw[0] |= m; else e.capw[1] |= m;
}

__device__ __forceinline__ void send_home(Env &e, const Ctx &c, int i)
{   // capture.py:691-693 / 699-701 / 717-719 / 725-727
    e.pac[i] = 0; e.xy[i] = c.start_xy[i]; e.dir[i] = 4; e.scared[i] = 0;
}

// capture.py:569-668 dumpFoodFromDeath.  The reference's FIFO BFS visits offsets in a board-independent order;
// c.dump holds that order (generated on the host by running the same BFS), so the walk is a linear scan.
__device__ __forceinline__ void dump_food(Env &e, const Ctx &c, uint32_t who_xy, int num, int &d_red, int &d_blue)
{
    const int who_x = who_xy & 0xFF, who_y = who_xy >> 8;
    const int side_red = 2 * who_x < c.W;
    // Three things keep this rare path short (a tick in which some env dumps food used to be 4 us slower than one without,
    // and with 16 k envs in lock-step most ticks have one): (1) the table index is wave-uniform, so the entries come through
    // the scalar cache (constant address space) -- a vector load costs an L2 round trip per candidate AND waits for the
    // snapshot stores in front of it (stores share vmcnt with loads on gfx9); (2) every test of capture.py:604-629 except
    // "inside the rows" is folded once into per-row masks of cells that cannot take a pellet (LDS scratch column fd3), so a
    // candidate costs one row read and one bit test; (3) candidates are taken four at a time (one LDS round trip per group)
    // and judged strictly in order; a pellet is placed with an LDS OR (a cell occurs once in the visit order).
    typedef const uint32_t __attribute__((address_space(4))) *const_words_t;
    const const_words_t dwords = (const_words_t)(uintptr_t)c.dump;
    // rows of cells that cannot take a pellet: walls, food, the other side (2x < W decides, :618), the border columns x = 0 and
    // x >= W (:609), capsules (:621) and agents (:625-627); row 0 and rows >= H are rejected by the index test below
    const uint32_t inner = (c.W >= 32 ? 0xFFFFFFFFu : ((1u << c.W) - 1u)) & ~1u;
    const uint32_t red_cols = (1u << ((c.W + 1) >> 1)) - 1u;                  // x with 2x < W
    const uint32_t allowed = inner & (side_red ? red_cols : ~red_cols);
    c.fd3[0] = 0xFFFFFFFFu;
    for (int y = 1; y < c.H; ++y) c.fd3[y * PMX_RULE_BLOCK] = c.wl[y * c.wls] | c.fd[y * PMX_RULE_BLOCK] | ~allowed;
#pragma unroll
    for (int i = 0; i < 4; ++i) atomicOr(&c.fd3[(e.xy[i] >> 8) * PMX_RULE_BLOCK], 1u << (e.xy[i] & 0xFF));
    if ((e.capw[0] & e.capw[1]) != 0xFFFFFFFFu) {
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
            const uint32_t cxy = ((sl < 2 ? e.capw[0] : e.capw[1]) >> (16 * (sl & 1))) & 0xFFFFu;
            if (cxy != 0xFFFFu) atomicOr(&c.fd3[(cxy >> 8) * PMX_RULE_BLOCK], 1u << (cxy & 0xFF));
        }
    }
    for (int k = 0; k < c.n_dump && num > 0; k += 4) {
        const uint32_t w0 = dwords[k >> 1], w1 = dwords[(k >> 1) + 1];      // the table is padded with out-of-board offsets
        int X[4], Y[4];
        bool inb[4];
        uint32_t brow[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t ent = ((j < 2 ? w0 : w1) >> (16 * (j & 1))) & 0xFFFFu;
            X[j] = who_x + (int)(int8_t)(ent & 0xFF); Y[j] = who_y + (int)(int8_t)(ent >> 8);
            inb[j] = (uint32_t)Y[j] < (uint32_t)c.H && (uint32_t)X[j] < 32u;
            brow[j] = c.fd3[(inb[j] ? Y[j] : 0) * PMX_RULE_BLOCK];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (num <= 0 || !inb[j] || ((brow[j] >> X[j]) & 1u)) continue;
            atomicOr(&c.fd[Y[j] * PMX_RULE_BLOCK], 1u << X[j]);
            --num;
            if (X[j] < c.half) ++d_red; else ++d_blue;
        }
    }
}

template <int WHO>
__device__ __forceinline__ void kill_dump(Env &e, const Ctx &c, int &d_red, int &d_blue)
{
    if (e.pac[WHO] && e.carry[WHO] > 0) dump_food(e, c, e.xy[WHO], e.carry[WHO], d_red, d_blue);
    if (e.pac[WHO]) e.carry[WHO] = 0;   // (a non-Pacman here is the reference's "seriously wrong" raise: unreachable)
}

// capture.py:519-560 consume, for an eater of team RED / blue standing on (px, py)
template <bool RED>
__device__ __forceinline__ void consume(Env &e, const Ctx &c, int px, int py, int &d_red, int &d_blue)
{
    const uint32_t key = (uint32_t)px | ((uint32_t)py << 8);
    uint32_t row = c.fd[py * PMX_RULE_BLOCK];
    if ((row >> px) & 1u) {
        constexpr int T1 = RED ? 0 : 1, T2 = T1 + 2;                      // :533-537 team order
        if (e.xy[T1] == key) e.carry[T1] += 1;
        else if (e.xy[T2] == key) e.carry[T2] += 1;
        c.fd[py * PMX_RULE_BLOCK] = row & ~(1u << px);
        if (px < c.half) --d_red; else --d_blue;
    }
    int slot = cap_find(e, key);
    if (slot >= 0) {
        bool mine = RED ? (2 * px > c.W) : (2 * px <= c.W);               // halfList, capture.py:344-350
        if (mine) {
            cap_remove(e, slot);
            constexpr int O1 = RED ? 1 : 0, O2 = O1 + 2;
            e.scared[O1] = PMX_SCARED_TIME; e.scared[O2] = PMX_SCARED_TIME;
        }
    }
}

// capture.py:107-123 generateSuccessor for mover I, in place.  Returns scoreChange; d_red/d_blue receive the net
// change of the food counts on the red / blue side (what gymPacMan.get_reward compares, gymPacMan.py:234-247).
// PMX_ACTION_RANDOM_LEGAL: uniform choice among the legal actions in the reference's list order N,S,E,W,Stop
// (agents/randomTeam.py:100 random.choice(actions)); lowbias32 hash of (seed, env, tick, agent) as the generator.
__device__ __forceinline__ int random_legal(int legal, uint32_t key, uint32_t ticks, int agent)
{
    uint32_t x = key ^ (ticks * 0x85EBCA77u) ^ ((uint32_t)agent * 0xC2B2AE3Du);
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    const uint32_t n = (uint32_t)__popc(legal);
    int k = (int)(((uint64_t)x * n) >> 32);
    int pick = 4;
    const int order[5] = { 0, 2, 1, 3, 4 };
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int a = order[j];
        if ((legal >> a) & 1) { if (k == 0) pick = a; --k; }
    }
    return pick;
}

template <int I>
__device__ __forceinline__ int substep_core(Env &e, const Ctx &c, int action, bool &req_legal, int &d_red, int &d_blue)
{
    constexpr bool RED = (I % 2) == 0;
    constexpr int O1 = RED ? 1 : 0, O2 = O1 + 2;
    // ---- applyAction capture.py:468-517
    int px = e.xy[I] & 0xFF, py = e.xy[I] >> 8;
    const int legal = legal_mask(c.wl, c.wls, px, py);
    req_legal = (action >= 0) && (action <= 4) && ((legal >> (action & 7)) & 1);
    if (!req_legal) action = 4;                                           // :473-474
    px += (action == 1) - (action == 3); py += (action == 0) - (action == 2);
    e.xy[I] = (uint32_t)px | ((uint32_t)py << 8);
    if (action != 4) e.dir[I] = action;                                   // game.py:115-118
    e.pac[I] = (int)(RED != (2 * px < c.W));                              // :492
    int eater_pac = e.pac[I];
    int sc = 0;
    if (e.carry[I] > 0 && !e.pac[I]) {                                    // :495-511
        sc = RED ? e.carry[I] : -e.carry[I];
        e.ret[I] += e.carry[I];
        e.carry[I] = 0;
        eater_pac = e.pac[3];                                             // :505 leaves agentState bound to agent 3
    }
    if (eater_pac) consume<RED>(e, c, px, py, d_red, d_blue);            // :514-515
    // ---- checkDeath capture.py:670-728.  Nothing happens unless an opponent stands on the mover's cell, so the common
    // case is two compares; the sequential logic (the isPacman branch is chosen once, positions are re-read per
    // opponent because the mover may have been sent home in between) runs only on a collision.
    if (e.xy[O1] == e.xy[I] || e.xy[O2] == e.xy[I]) {
        if (e.pac[I]) {
            if (!e.pac[O1] && e.xy[O1] == e.xy[I]) {
                if (e.scared[O1] <= 0) { kill_dump<I>(e, c, d_red, d_blue); send_home(e, c, I); }
                else send_home(e, c, O1);
            }
            if (!e.pac[O2] && e.xy[O2] == e.xy[I]) {
                if (e.scared[O2] <= 0) { kill_dump<I>(e, c, d_red, d_blue); send_home(e, c, I); }
                else send_home(e, c, O2);
            }
        } else {
            if (e.pac[O1] && e.xy[O1] == e.xy[I]) {
                if (e.scared[I] <= 0) { kill_dump<O1>(e, c, d_red, d_blue); send_home(e, c, O1); }
                else send_home(e, c, I);
            }
            if (e.pac[O2] && e.xy[O2] == e.xy[I]) {
                if (e.scared[I] <= 0) { kill_dump<O2>(e, c, d_red, d_blue); send_home(e, c, O2); }
                else send_home(e, c, I);
            }
        }
    }
    e.scared[I] = e.scared[I] > 1 ? e.scared[I] - 1 : 0;                  // capture.py:562-567, mover only
    e.score += sc;                                                        // capture.py:121
    return sc;
}

// agents/baselineTeam.py:65-187 evaluated in the kernel (action codes -3 offensive / -4 defensive, include/pmx.h): the
// successor of every legal action is generated for real (on a register copy of the agents and a scratch LDS copy of the
// food column), scored with the reference's features x weights, and one of the best actions is drawn with the
// counter-based generator; with no food left to eat the agent walks home (:81-90).
template <int I>
__device__ __noinline__ int bot_action(const Env &e, const Ctx &c, bool defensive)
{
    constexpr bool RED = (I % 2) == 0;
    constexpr int O1 = RED ? 1 : 0, O2 = O1 + 2;
    const int legal = legal_mask(c.wl, c.wls, (int)(e.xy[I] & 0xFF), (int)(e.xy[I] >> 8));
    const uint32_t enemy_mask = RED ? c.hi_mask : c.lo_mask;      // getFood: the other side's pellets
    int food_left = 0;
    for (int y = 0; y < c.H; ++y) food_left += __popc(c.fd[y * PMX_RULE_BLOCK] & enemy_mask);
    const int start_idx = c.cidx[(c.start_xy[I] >> 8) * 32 + (c.start_xy[I] & 0xFF)];
    int best = -(1 << 30), best_mask = 0, home_best = 9999, home_act = -1;
    const int order[5] = { 0, 2, 1, 3, 4 };
    const int rev[5] = { 2, 3, 0, 1, 4 };
    for (int k = 0; k < 5; ++k) {
        const int a = order[k];
        if (!((legal >> a) & 1)) continue;
        Env t = e;
        for (int y = 0; y < c.H; ++y) c.fd2[y * PMX_RULE_BLOCK] = c.fd[y * PMX_RULE_BLOCK];
        Ctx c2 = c;
        c2.fd = c.fd2;
        bool rl; int dr = 0, db = 0;
        substep_core<I>(t, c2, a, rl, dr, db);
        const int my = c.cidx[(t.xy[I] >> 8) * 32 + (t.xy[I] & 0xFF)];
        const uint8_t *drow = c.dist + (size_t)my * c.n_cells;
        int val;
        if (!defensive) {
            int cnt = 0, mind = 1 << 30;
            for (int y = 0; y < c.H; ++y) {
                uint32_t m = c.fd2[y * PMX_RULE_BLOCK] & enemy_mask;
                cnt += __popc(m);
                while (m) {
                    const int x = __ffs(m) - 1;
                    m &= m - 1;
                    const int d = drow[c.cidx[y * 32 + x]];
                    mind = d < mind ? d : mind;
                }
            }
            val = -100 * cnt - (cnt > 0 ? mind : 0);
        } else {
            int num_inv = 0, mind = 1 << 30;
            if (t.pac[O1]) { ++num_inv; const int d = drow[c.cidx[(t.xy[O1] >> 8) * 32 + (t.xy[O1] & 0xFF)]]; mind = d < mind ? d : mind; }
            if (t.pac[O2]) { ++num_inv; const int d = drow[c.cidx[(t.xy[O2] >> 8) * 32 + (t.xy[O2] & 0xFF)]]; mind = d < mind ? d : mind; }
            val = -1000 * num_inv + 100 * (t.pac[I] ? 0 : 1) - (num_inv > 0 ? 10 * mind : 0) - (a == 4 ? 100 : 0) - (a == rev[e.dir[I]] ? 2 : 0);
        }
        if (val > best) { best = val; best_mask = 1 << a; }
        else if (val == best) best_mask |= 1 << a;
        const int hd = c.dist[(size_t)start_idx * c.n_cells + my];
        if (hd < home_best) { home_best = hd; home_act = a; }
    }
    if (food_left <= 0) return home_act;
    uint32_t x = c.rng_key ^ (e.ticks * 0x85EBCA77u) ^ ((uint32_t)I * 0xC2B2AE3Du) ^ 0x5bd1e995u;
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    int kk = (int)(((uint64_t)x * (uint32_t)__popc(best_mask)) >> 32);
    int pick = 4;
    for (int j = 0; j < 5; ++j) {
        const int a = order[j];
        if ((best_mask >> a) & 1) { if (kk == 0) pick = a; --kk; }
    }
    return pick;
}

// BOTS: the handle was created with pmx_config.enable_bots; only that kernel variant carries the reflex-bot code (it costs
// registers and a stack frame for the agent state: +3 us per tick on 16 k envs even when no bot code is used)
template <int I, bool BOTS>
__device__ __forceinline__ int substep(Env &e, const Ctx &c, int action, bool &req_legal, int &d_red, int &d_blue)
{
    if (action == -2) action = random_legal(legal_mask(c.wl, c.wls, (int)(e.xy[I] & 0xFF), (int)(e.xy[I] >> 8)), c.rng_key, e.ticks, I);
    if constexpr (BOTS) {
        if ((action == -3 || action == -4) && c.dist) action = bot_action<I>(e, c, action == -4);
    }
    return substep_core<I>(e, c, action, req_legal, d_red, d_blue);
}

// One iteration of the loop gymPacMan.py:149-169 for agent I: shaped reward (from the successor, which is what the
// reference's shadow successor equals) + the transition itself.
template <int I, bool BOTS>
__device__ __forceinline__ void tick_substep(Env &e, Acc &a, const Ctx &c, int action)
{
    constexpr bool RED = (I % 2) == 0;
    constexpr int O1 = RED ? 1 : 0, O2 = O1 + 2;
    const int pac1 = e.pac[O1], pac2 = e.pac[O2];
    bool req_legal;
    int d_red = 0, d_blue = 0;
    const int sc = substep<I, BOTS>(e, c, action, req_legal, d_red, d_blue);
    double r = RED ? a.red_r : a.blue_r;
    if (RED) {                                                            // gymPacMan.py:233-242
        if (d_red > 0) r += 1.0;
        if (d_blue < 0) r += 0.1;
    } else {                                                              // :244-253
        if (d_blue > 0) r += 1.0;
        if (d_red < 0) r += 0.1;
    }
    if (c.defence_reward) {
        if (pac1 && !e.pac[O1] && e.xy[O1] == c.start_xy[O1]) r += 0.25;
        if (pac2 && !e.pac[O2] && e.xy[O2] == c.start_xy[O2]) r += 0.25;
    }
    if (c.legal_reward && req_legal) r += 0.01;                           // :254-257
    if (RED) { a.red_r = r; a.red_sc += sc; } else { a.blue_r = r; a.blue_sc -= sc; }
    a.n_red += d_red; a.n_blue += d_blue;
    a.sc_total += sc;                                                     // :153-162
    e.self_after[I] = e.xy[I] | ((uint32_t)e.carry[I] << 16);
}

// HB > 0: the board has at most HB rows and the row loops are fully unrolled with predicated accesses, so that all HBM
// loads (or all LDS reads) of a state are in flight at once.  With a run-time trip count the compiler emits
// load -> s_waitcnt -> ds_write per row: H dependent memory round trips on a kernel that runs one wave per SIMD.
// The run-time loops remain for the per-agent and query kernels, which are not on the hot path.
struct RawEnv {            // the non-row state words as loaded, before unpacking
    uint32_t a[4], b[4], capw[2], score, steps, ticks;
};
// phase 1 of the hot-path load: nothing but global loads (needs only the kernel arguments), issued before the layout
// set-up so that the HBM latency overlaps it
template <int HB>
__device__ __forceinline__ void load_env_issue(RawEnv &r, uint32_t (&rows)[HB], const uint32_t *st, int N, int env, int H)
{
#pragma unroll
    for (int y = 0; y < HB; ++y) rows[y] = y < H ? st[(size_t)y * N + env] : 0u;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        r.a[i] = st[(size_t)PMX_W_AGENT_A(H, i) * N + env];
        r.b[i] = st[(size_t)PMX_W_AGENT_B(H, i) * N + env];
    }
    r.capw[0] = st[(size_t)PMX_W_CAPS(H, 0) * N + env];
    r.capw[1] = st[(size_t)PMX_W_CAPS(H, 1) * N + env];
    r.score = st[(size_t)PMX_W_SCORE(H) * N + env];
    r.steps = st[(size_t)PMX_W_STEPS(H) * N + env];
    r.ticks = st[(size_t)PMX_W_TICKS(H) * N + env];
}
// phase 2: unpack into registers, food rows into this lane's LDS column
template <int HB>
__device__ __forceinline__ void load_env_commit(Env &e, const Ctx &c, const RawEnv &r, const uint32_t (&rows)[HB])
{
#pragma unroll
    for (int i = 0; i < 4; ++i) { unpack_a(e, i, r.a[i]); unpack_b(e, i, r.b[i]); }
    e.capw[0] = r.capw[0]; e.capw[1] = r.capw[1];
    e.score = (int)r.score; e.steps = (int)r.steps; e.ticks = r.ticks;
#pragma unroll
    for (int y = 0; y < HB; ++y)
        if (y < c.H) c.fd[y * PMX_RULE_BLOCK] = rows[y];
}
// capture.py:332-342 halfGrid sums of the state as loaded (the rows are still in registers: no LDS round trip)
template <int HB>
__device__ __forceinline__ void count_food(Acc &a, const Ctx &c, const uint32_t (&rows)[HB])
{
    int nr = 0, nb = 0;
#pragma unroll
    for (int y = 0; y < HB; ++y) { nr += __popc(rows[y] & c.lo_mask); nb += __popc(rows[y] & c.hi_mask); }
    a.n_red = nr; a.n_blue = nb;
}

__device__ __forceinline__ void load_env(Env &e, const Ctx &c, const uint32_t *st, int N, int env)
{
    for (int y = 0; y < c.H; ++y) c.fd[y * PMX_RULE_BLOCK] = st[(size_t)y * N + env];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unpack_a(e, i, st[(size_t)PMX_W_AGENT_A(c.H, i) * N + env]);
        unpack_b(e, i, st[(size_t)PMX_W_AGENT_B(c.H, i) * N + env]);
    }
    e.capw[0] = st[(size_t)PMX_W_CAPS(c.H, 0) * N + env];
    e.capw[1] = st[(size_t)PMX_W_CAPS(c.H, 1) * N + env];
    e.score = (int)st[(size_t)PMX_W_SCORE(c.H) * N + env];
    e.steps = (int)st[(size_t)PMX_W_STEPS(c.H) * N + env];
    e.ticks = st[(size_t)PMX_W_TICKS(c.H) * N + env];
}
template <int HB = 0>
__device__ __forceinline__ void store_snapshot(const Env &e, const Ctx &c, uint32_t *st, int N, int env)
{
    if constexpr (HB > 0) {
        uint32_t rows[HB];
#pragma unroll
        for (int y = 0; y < HB; ++y) rows[y] = y < c.H ? c.fd[y * PMX_RULE_BLOCK] : 0u;
#pragma unroll
        for (int y = 0; y < HB; ++y)
            if (y < c.H) st[(size_t)y * N + env] = rows[y];
    } else {
        for (int y = 0; y < c.H; ++y) st[(size_t)y * N + env] = c.fd[y * PMX_RULE_BLOCK];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        st[(size_t)PMX_W_AGENT_A(c.H, i) * N + env] = pack_a(e, i);
        st[(size_t)PMX_W_AGENT_B(c.H, i) * N + env] = pack_b(e, i);
    }
    st[(size_t)PMX_W_CAPS(c.H, 0) * N + env] = e.capw[0];
    st[(size_t)PMX_W_CAPS(c.H, 1) * N + env] = e.capw[1];
}
// The same words with 16-byte stores: a dword store per word and lane is 22 (snapshot) or 25 (state) store instructions of
// 256 bytes each, and on a kernel that runs ONE wave per CU their issue time is on the critical path (s_memtime: a sub-step
// with its snapshot took 2 250-2 700 ticks, one without 1 200).  The SoA rows of 64 consecutive envs are 256 contiguous
// bytes, and the food rows already sit in LDS as [row][lane]: lane L of instruction k reads row 4k + (L >> 4), envs
// 4 (L & 15) .. + 3 with one ds_read_b128 and stores them with one global_store_dwordx4 -- 3 + 3 instructions per snapshot
// instead of 22.  The non-row words go through an LDS staging block in the same shape.  Needs all 64 lanes live
// (N % 64 == 0), which the caller checks; LDS operations of one wave execute in order, so later row updates cannot overtake.
// rows 4k + sub (k = K0 .. K1 - 1) of an LDS block [row][lane] -> SoA words (word0 + row) of 64 envs, 16 bytes per lane
template <int K0, int K1>
__device__ __forceinline__ void wide_group(const uint32_t *lds_rows, int n_rows, int word0, int sub, int q4, uint32_t *dst, int N)
{
    if constexpr (K0 < K1) {
        const int r = 4 * K0 + sub;
        const uint4 v = *reinterpret_cast<const uint4 *>(lds_rows + (r < n_rows ? r : 0) * PMX_RULE_BLOCK + q4);
        wide_group<K0 + 1, K1>(lds_rows, n_rows, word0, sub, q4, dst, N);       // the later reads are issued before this store
        if (r < n_rows) *reinterpret_cast<uint4 *>(dst + (size_t)(word0 + r) * N) = v;
    }
}
template <int HB, int NW>
__device__ __forceinline__ void store_words_wide(const Ctx &c, const uint32_t (&w)[NW], uint32_t *st, int N)
{
    const int lane = threadIdx.x, sub = lane >> 4, q4 = 4 * (lane & 15);
    uint32_t *dst = st + (size_t)blockIdx.x * PMX_RULE_BLOCK + q4;
#pragma unroll
    for (int i = 0; i < NW; ++i) c.stg[i * PMX_RULE_BLOCK + lane] = w[i];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // (reads unconditional, with the row clamped, and no register ARRAY of them: an array the compiler cannot fully scalarise
    // goes to scratch memory, and a scratch reload waits for vmcnt(0), i.e. for every store in flight)
    constexpr int KR = HB / 4, KW = (NW + 3) / 4;
    wide_group<0, KR>(c.rows0, c.H, 0, sub, q4, dst, N);
    wide_group<0, KW>(c.stg, NW, c.H, sub, q4, dst, N);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");          // the staging block is rewritten by the next call
    __builtin_amdgcn_wave_barrier();
}
template <int HB>
__device__ __forceinline__ void store_snapshot_wide(const Env &e, const Ctx &c, uint32_t *st, int N)
{
    uint32_t w[10];
#pragma unroll
    for (int i = 0; i < 4; ++i) { w[i] = pack_a(e, i); w[4 + i] = pack_b(e, i); }
    w[8] = e.capw[0]; w[9] = e.capw[1];
    store_words_wide<HB, 10>(c, w, st, N);
}
template <int HB>
__device__ __forceinline__ void store_env_wide(const Env &e, const Ctx &c, uint32_t *st, int N)
{
    uint32_t w[13];
#pragma unroll
    for (int i = 0; i < 4; ++i) { w[i] = pack_a(e, i); w[4 + i] = pack_b(e, i); }
    w[8] = e.capw[0]; w[9] = e.capw[1];
    w[10] = (uint32_t)e.score; w[11] = (uint32_t)e.steps; w[12] = e.ticks;
    store_words_wide<HB, 13>(c, w, st, N);
}
template <int HB = 0>
__device__ __forceinline__ void store_env(const Env &e, const Ctx &c, uint32_t *st, int N, int env)
{
    store_snapshot<HB>(e, c, st, N, env);
    st[(size_t)PMX_W_SCORE(c.H) * N + env] = (uint32_t)e.score;
    st[(size_t)PMX_W_STEPS(c.H) * N + env] = (uint32_t)e.steps;
    st[(size_t)PMX_W_TICKS(c.H) * N + env] = e.ticks;
}
// random_layout=True (gymPacMan.py:98-100): the layout an env moves to when it is reset; counter-based draw keyed by
// (seed, env, the env's tick counter) -- a build-side definition, include/pmx.h redraw_layouts
__device__ __forceinline__ int redraw_layout(uint32_t key, uint32_t ticks, int n_layouts)
{
    uint32_t x = key ^ (ticks * 0x85EBCA77u) ^ 0x4C41594Fu;
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return (int)(((uint64_t)x * (uint32_t)n_layouts) >> 32);
}

// point the lane's context at another layout of the pool: layout record, start cells, and the lane's wall column in LDS
__device__ __forceinline__ void switch_layout(Ctx &c, const PmxTickParams &p, int env, int li)
{
    p.layout_idx_rw[env] = li;
    c.L = p.lay + li;
#pragma unroll
    for (int i = 0; i < 4; ++i) c.start_xy[i] = (uint32_t)c.L->startx[i] | ((uint32_t)c.L->starty[i] << 8);
    uint32_t *w = const_cast<uint32_t *>(c.wl);
    for (int y = 0; y < 32; ++y) w[y * c.wls] = y < c.H ? c.L->walls[y] : 0xFFFFFFFFu;
}

__device__ __forceinline__ void init_env(Env &e, const Ctx &c)
{
    for (int y = 0; y < c.H; ++y) c.fd[y * PMX_RULE_BLOCK] = c.L->food0[y];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        e.xy[i] = c.start_xy[i]; e.dir[i] = 4;
        e.pac[i] = 0; e.scared[i] = 0; e.carry[i] = 0; e.ret[i] = 0;
    }
    e.capw[0] = c.L->capw0[0]; e.capw[1] = c.L->capw0[1];
    e.score = 0; e.steps = 0;
}

// gymPacMan.py:171-193: everything after the four sub-steps.
template <int HB = 0>
__device__ __forceinline__ void tick_finish(Env &e, Acc &a, const Ctx &c_in, const PmxTickParams &p, int env, bool fused)
{
    Ctx c = c_in;             // a reset may move the env to another layout of the pool (redraw_layouts)
    double blue_r = a.blue_r + (double)(a.blue_sc > 0 ? a.blue_sc : 0);   // :171-172
    double red_r = a.red_r + (double)(a.red_sc > 0 ? a.red_sc : 0);
    int n_red = 0, n_blue = 0;                                            // capture.py:332-342 halfGrid sums
    if constexpr (HB > 0) {
        n_red = a.n_red; n_blue = a.n_blue;      // counted when the state was loaded, updated by every eaten / dumped pellet
    } else {
        for (int y = 0; y < c.H; ++y) {
            uint32_t row = c.fd[y * PMX_RULE_BLOCK];
            n_red += __popc(row & c.lo_mask);
            n_blue += __popc(row & c.hi_mask);
        }
    }
    bool done = (n_blue == 0 && e.carry[0] == 0 && e.carry[2] == 0) ||    // gymPacMan.py:261-270
                (n_red == 0 && e.carry[1] == 0 && e.carry[3] == 0) || (e.steps >= p.length);
    if (done) {                                                           // :177-182
        const int fs = e.score;
        if (fs < 0) blue_r += 20.0 + (1.0 / 5) * (double)(-fs);
        else if (fs > 0) red_r += 20.0 + (1.0 / 5) * (double)fs;
    }
    e.steps += 1;                                                         // :189
    e.ticks += 1;
    if (p.agent_out && fused) {
        uint4 v = make_uint4(e.self_after[0], e.self_after[1], e.self_after[2], e.self_after[3]);
        reinterpret_cast<uint4 *>(p.agent_out)[env] = v;
    }
    if (p.reward) { p.reward[2 * (size_t)env] = red_r; p.reward[2 * (size_t)env + 1] = blue_r; }
    if (p.done) p.done[env] = (uint8_t)done;
    if (p.score_change) p.score_change[env] = a.sc_total;
    if (p.score) p.score[env] = e.score;
    if (done && p.auto_reset) {
        if (p.layout_idx_rw) switch_layout(c, p, env, redraw_layout(c.rng_key, e.ticks, p.n_layouts));
        init_env(e, c);
        // the observations of a finished env are those of the fresh game for all four agents (gymPacMan.py:135-137)
        const size_t snap_sz = (size_t)PMX_SNAP_WORDS(c.H) * p.N;
        for (int s = 0; s < 3; ++s) store_snapshot(e, c, p.snap + s * snap_sz, p.N, env);
    }
    if (p.legal) {
        uint32_t m = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) m |= (uint32_t)legal_mask(c.wl, c.wls, (int)(e.xy[i] & 0xFF), (int)(e.xy[i] >> 8)) << (8 * i);
        reinterpret_cast<uint32_t *>(p.legal)[env] = m;
    }
}

__device__ __forceinline__ Ctx make_ctx(const PmxTickParams &p, uint32_t *lds)
{
    Ctx c;
    const int env0 = blockIdx.x * PMX_RULE_BLOCK + threadIdx.x;
    const bool multi = p.layout_idx != nullptr;
    c.L = p.lay + ((multi && env0 < p.N) ? p.layout_idx[env0] : 0);
    c.W = p.lay_W; c.H = p.lay_H; c.half = p.lay_half; c.n_dump = p.lay_n_dump;       // identical in every layout of a handle
    c.dump = p.dump;
    c.legal_reward = p.legal_reward; c.defence_reward = p.defence_reward;
    c.wl = multi ? lds + 32 + PMX_MAX_H_LDS * PMX_RULE_BLOCK + threadIdx.x : lds;
    c.wls = multi ? PMX_RULE_BLOCK : 1;
    c.fd = lds + 32 + threadIdx.x;
    c.rows0 = lds + 32;
    c.stg = lds + 32 + (multi ? (3 * PMX_MAX_H_LDS + 32) : 3 * c.H) * PMX_RULE_BLOCK;
    c.fd2 = multi ? lds + 32 + 2 * PMX_MAX_H_LDS * PMX_RULE_BLOCK + threadIdx.x : lds + 32 + c.H * PMX_RULE_BLOCK + threadIdx.x;
    c.fd3 = multi ? lds + 32 + 3 * PMX_MAX_H_LDS * PMX_RULE_BLOCK + threadIdx.x : lds + 32 + 2 * c.H * PMX_RULE_BLOCK + threadIdx.x;
    c.dist = p.dist ? p.dist + c.L->dist_off : nullptr;
    c.cidx = p.cell_index ? p.cell_index + (size_t)(c.L - p.lay) * 1024 : nullptr;
    c.n_cells = c.L->n_cells;
    c.lo_mask = p.lo_mask; c.hi_mask = p.hi_mask;
#pragma unroll
    for (int i = 0; i < 4; ++i) c.start_xy[i] = (uint32_t)c.L->startx[i] | ((uint32_t)c.L->starty[i] << 8);
    c.rng_key = p.seed ^ ((uint32_t)(blockIdx.x * PMX_RULE_BLOCK + threadIdx.x) * 0x9E3779B1u);
    if (threadIdx.x < 32) lds[threadIdx.x] = threadIdx.x < (unsigned)c.H ? p.lay->walls[threadIdx.x] : 0xFFFFFFFFu;
    if (multi) {   // per-env layouts: every lane keeps its own wall column next to its food column
        uint32_t *w = lds + 32 + PMX_MAX_H_LDS * PMX_RULE_BLOCK + threadIdx.x;
        uint32_t wr[32];
#pragma unroll
        for (int y = 0; y < 32; ++y) wr[y] = y < c.H ? c.L->walls[y] : 0xFFFFFFFFu;     // all loads in flight, then the LDS writes
#pragma unroll
        for (int y = 0; y < 32; ++y) w[y * PMX_RULE_BLOCK] = wr[y];
    }
    __syncthreads();
    return c;
}

}  // namespace

// Development aid (-DPMX_RULE_TIMING, never in the shipped build): s_memtime stamps of the phases of the wave of block 0, summed
// over launches in pmx_rule_ticks[] (start->ctx, ->state unpacked, four sub-steps with their snapshot stores, finish, state store)
#ifdef PMX_RULE_TIMING
__device__ unsigned long long pmx_rule_ticks[16];
#define PMX_RTICK(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); if (blockIdx.x == 0 && threadIdx.x == 0) pmx_rule_ticks[i] += now_ - last_; last_ = now_; } while (0)
#else
#define PMX_RTICK(i) do { } while (0)
#endif

// dynamic LDS: 32 wall rows + 3 x H rows x PMX_RULE_BLOCK lanes (food, bots' scratch copy, dump_food's blocked-cell rows)
template <bool BOTS, int HB>
__global__ __launch_bounds__(PMX_RULE_BLOCK) void pmx_rule_kernel(PmxTickParams p)
{
    extern __shared__ uint32_t lds[];
#ifdef PMX_RULE_TIMING
    unsigned long long last_ = __builtin_amdgcn_s_memtime();
#endif
    const int env = blockIdx.x * PMX_RULE_BLOCK + threadIdx.x;
    const bool live = env < p.N;
    RawEnv raw;
    uint32_t rows[HB];
    uint32_t av = 0;
    if (live) {
        load_env_issue<HB>(raw, rows, p.state, p.N, env, p.lay_H);
        av = reinterpret_cast<const uint32_t *>(p.actions)[env];   // 4 int8 actions
    }
    Ctx c = make_ctx(p, lds);
    PMX_RTICK(0);
    if (!live) return;
    Env e;
    load_env_commit<HB>(e, c, raw, rows);
    PMX_RTICK(1);
    Acc a = { 0.0, 0.0, 0, 0, 0, 0, 0 };
    count_food<HB>(a, c, rows);
    const size_t snap_sz = (size_t)PMX_SNAP_WORDS(c.H) * p.N;
    const bool wide = (p.N & (PMX_RULE_BLOCK - 1)) == 0;          // every wave is full: the 16-byte store path (store_words_wide)
    tick_substep<0, BOTS>(e, a, c, (int)(int8_t)(av & 0xFF));
    if (wide) store_snapshot_wide<HB>(e, c, p.snap, p.N); else store_snapshot<HB>(e, c, p.snap, p.N, env);
    PMX_RTICK(2);
    tick_substep<1, BOTS>(e, a, c, (int)(int8_t)((av >> 8) & 0xFF));
    if (wide) store_snapshot_wide<HB>(e, c, p.snap + snap_sz, p.N); else store_snapshot<HB>(e, c, p.snap + snap_sz, p.N, env);
    PMX_RTICK(3);
    tick_substep<2, BOTS>(e, a, c, (int)(int8_t)((av >> 16) & 0xFF));
    if (wide) store_snapshot_wide<HB>(e, c, p.snap + 2 * snap_sz, p.N); else store_snapshot<HB>(e, c, p.snap + 2 * snap_sz, p.N, env);
    PMX_RTICK(4);
    tick_substep<3, BOTS>(e, a, c, (int)(int8_t)((av >> 24) & 0xFF));
    PMX_RTICK(5);
    tick_finish<HB>(e, a, c, p, env, true);
    PMX_RTICK(6);
    if (wide) store_env_wide<HB>(e, c, p.state, p.N); else store_env<HB>(e, c, p.state, p.N, env);
    PMX_RTICK(7);
#ifdef PMX_RULE_TIMING
    if (blockIdx.x == 0 && threadIdx.x == 0) pmx_rule_ticks[15] += 1;
#endif
}

// pmx_step_agent: one sub-step; the accumulators of the open tick travel through the state words.
template <int I, bool BOTS>
__device__ __forceinline__ void rule_agent_body(const PmxTickParams &p, uint32_t *lds)
{
    Ctx c = make_ctx(p, lds);
    const int env = blockIdx.x * PMX_RULE_BLOCK + threadIdx.x;
    if (env >= p.N) return;
    Env e;
    load_env(e, c, p.state, p.N, env);
    Acc a = { 0.0, 0.0, 0, 0, 0, 0, 0 };
    uint32_t *acc = p.state + (size_t)PMX_W_ACC(c.H) * p.N + env;
    if (I > 0) {
        a.red_r = __hiloint2double((int)acc[(size_t)1 * p.N], (int)acc[0]);
        a.blue_r = __hiloint2double((int)acc[(size_t)3 * p.N], (int)acc[(size_t)2 * p.N]);
        a.red_sc = (int)acc[(size_t)4 * p.N]; a.blue_sc = (int)acc[(size_t)5 * p.N]; a.sc_total = (int)acc[(size_t)6 * p.N];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)   // sub-steps not yet taken report the current state
        e.self_after[i] = e.xy[i] | ((uint32_t)e.carry[i] << 16);
    tick_substep<I, BOTS>(e, a, c, (int)p.actions[env]);
    if (p.agent_out) p.agent_out[4 * (size_t)env + I] = e.self_after[I];
    if (I == 3) {
        tick_finish(e, a, c, p, env, false);
    } else {
        acc[0] = (uint32_t)__double2loint(a.red_r); acc[(size_t)1 * p.N] = (uint32_t)__double2hiint(a.red_r);
        acc[(size_t)2 * p.N] = (uint32_t)__double2loint(a.blue_r); acc[(size_t)3 * p.N] = (uint32_t)__double2hiint(a.blue_r);
        acc[(size_t)4 * p.N] = (uint32_t)a.red_sc; acc[(size_t)5 * p.N] = (uint32_t)a.blue_sc; acc[(size_t)6 * p.N] = (uint32_t)a.sc_total;
    }
    store_env(e, c, p.state, p.N, env);
}

// GameState.generateSuccessor as a stand-alone query (pmx_successor)
template <int I>
__device__ __forceinline__ void successor_body(const PmxTickParams &p, uint32_t *lds)
{
    Ctx c = make_ctx(p, lds);
    const int env = blockIdx.x * PMX_RULE_BLOCK + threadIdx.x;
    if (env >= p.N) return;
    Env e;
    load_env(e, c, p.state, p.N, env);
    bool req_legal;
    int d_red = 0, d_blue = 0;
    const int sc = substep<I, false>(e, c, (int)p.actions[env], req_legal, d_red, d_blue);
    if (p.score_change) p.score_change[env] = sc;
    store_env(e, c, p.state, p.N, env);
}

extern "C" __global__ __launch_bounds__(PMX_RULE_BLOCK) void pmx_successor_kernel(PmxTickParams p, int agent)
{
    extern __shared__ uint32_t lds[];
    switch (agent) {   // wave-uniform
    case 0: successor_body<0>(p, lds); break;
    case 1: successor_body<1>(p, lds); break;
    case 2: successor_body<2>(p, lds); break;
    default: successor_body<3>(p, lds); break;
    }
}

template <bool BOTS>
__global__ __launch_bounds__(PMX_RULE_BLOCK) void pmx_rule_agent_kernel(PmxTickParams p, int agent)
{
    extern __shared__ uint32_t lds[];
    switch (agent) {   // wave-uniform
    case 0: rule_agent_body<0, BOTS>(p, lds); break;
    case 1: rule_agent_body<1, BOTS>(p, lds); break;
    case 2: rule_agent_body<2, BOTS>(p, lds); break;
    default: rule_agent_body<3, BOTS>(p, lds); break;
    }
}

// gymPacMan.reset (gymPacMan.py:92-141) for the masked envs; legal masks for all envs if requested
extern "C" __global__ __launch_bounds__(PMX_RULE_BLOCK) void pmx_reset_kernel(PmxTickParams p)
{
    extern __shared__ uint32_t lds[];
    Ctx c = make_ctx(p, lds);
    const int env = blockIdx.x * PMX_RULE_BLOCK + threadIdx.x;
    if (env >= p.N) return;
    Env e;
    if (!p.no_reset && (!p.reset_mask || p.reset_mask[env])) {
        e.ticks = p.state[(size_t)PMX_W_TICKS(c.H) * p.N + env];
        if (p.layout_idx_rw) switch_layout(c, p, env, redraw_layout(c.rng_key, e.ticks, p.n_layouts));
        init_env(e, c);
        store_env(e, c, p.state, p.N, env);
        for (int k = 0; k < 7; ++k) p.state[(size_t)(PMX_W_ACC(c.H) + k) * p.N + env] = 0;
    } else if (p.legal) {
        load_env(e, c, p.state, p.N, env);
    }
    if (p.legal) {
        uint32_t m = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) m |= (uint32_t)legal_mask(c.wl, c.wls, (int)(e.xy[i] & 0xFF), (int)(e.xy[i] >> 8)) << (8 * i);
        reinterpret_cast<uint32_t *>(p.legal)[env] = m;
    }
}

template <int DT> struct ObsVec;
template <> struct ObsVec<0> { static constexpr int VEC = 4; };    // float32
template <> struct ObsVec<1> { static constexpr int VEC = 8; };    // bfloat16
template <> struct ObsVec<2> { static constexpr int VEC = 16; };   // uint8

// VEC stream bits (bit j = element j of the 16-byte vector) -> the 16 bytes.  Set elements are 1 in the output type.
template <int DT>
__device__ __forceinline__ uint4 pack_obs(uint32_t bits)
{
    uint32_t w[4];
    if (DT == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = (uint32_t)__builtin_amdgcn_sbfe((int)bits, j, 1) & 0x3F800000u;
    } else if (DT == 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            w[k] = ((uint32_t)__builtin_amdgcn_sbfe((int)bits, 2 * k, 1) & 0x3F80u) |
                   ((uint32_t)__builtin_amdgcn_sbfe((int)bits, 2 * k + 1, 1) & 0x3F800000u);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = (((bits >> (4 * k)) & 0xFu) * 0x00204081u) & 0x01010101u;   // bit j -> byte j
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// the one element of plane 1 that is set holds 1 + numCarrying (gymPacMan.py:205): patch element d of the vector
template <int DT>
__device__ __forceinline__ void patch_self(uint4 &v, int d, uint32_t carry)
{
    // uint8: selects on the four members, never an index into the vector: the compiler turns the unrolled loop of
    // `if (k == d >> 2) w[k] += ..` into a dynamically indexed access to a copy of the vector in SCRATCH memory, and every
    // scratch reload waits for vmcnt(0), i.e. for all the plane stores in flight (the uint8 kernels did exactly that; the
    // float32 / bfloat16 forms below compile to compares and selects and stay as they are)
    if (DT == 0) {
        uint32_t *w = reinterpret_cast<uint32_t *>(&v);
        const uint32_t val = __float_as_uint((float)(1 + carry));
#pragma unroll
        for (int j = 0; j < 4; ++j) if (d == j) w[j] = val;
    } else if (DT == 1) {
        uint32_t *w = reinterpret_cast<uint32_t *>(&v);
        const uint32_t val = __float_as_uint((float)(1 + carry)) >> 16;   // exact in bfloat16 for 1 + carry <= 256
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (d == 2 * k) w[k] = (w[k] & 0xFFFF0000u) | val;
            if (d == 2 * k + 1) w[k] = (w[k] & 0x0000FFFFu) | (val << 16);
        }
    } else {
        const uint32_t add = carry << (8 * (d & 3));
        const int k = d >> 2;
        v.x += k == 0 ? add : 0u; v.y += k == 1 ? add : 0u; v.z += k == 2 ? add : 0u; v.w += k == 3 ? add : 0u;
    }
}

// OR the W-bit row `v` into the packed bit stream at bit offset `off`
__device__ __forceinline__ void stream_or_row(uint32_t *T, uint32_t off, uint32_t v, int W)
{
    if (v == 0) return;
    const uint32_t word = off >> 5, sh = off & 31;
    atomicOr(&T[word], v << sh);
    if (sh + W > 32) atomicOr(&T[word + 1], v >> (32 - sh));
}

// ---------------------------------------------------------------------------------------------------------------
// Observation expansion (gymPacMan.py:195-229).  One wavefront per (env, emitted agent), wavefronts independent.
//   1. the wave assembles the PACKED bit stream of the [8][H][W] block in LDS (bit e = element e != 0): the wall
//      plane is a per-layout constant copied from the layout record, the food planes are OR-ed in row by row
//      (H lanes), self / ally / enemies / capsules are single bits (8 lanes);
//   2. every lane then reads ONE aligned stream word per 16 output bytes (a 4/8/16-bit field never straddles a
//      32-bit word), turns its bits into the 16 bytes through a block-shared look-up table and issues one dwordx4
//      store: the wave writes 1 KiB of consecutive addresses per instruction.  LDS operations of one wavefront
//      execute in order, so the waves need no barrier after the one that publishes the look-up table.
// ---------------------------------------------------------------------------------------------------------------
// LUT: the 16 output bytes of a lane come from a block-shared lookup table indexed by its stream bits (float32: 16 entries
// of 16 bytes, every entry in its own four LDS banks, so any mix of indices is conflict free; bfloat16: 256 entries; uint8: two
// 8-bit look-ups of 8 bytes) instead of being computed: with the observation buffer partly resident in the Infinity Cache the
// float32 kernel had become instruction bound (~1.8 T elements/s whatever the footprint), and the expansion arithmetic was most
// of its ~35 instructions per 16 bytes.
template <int DT, bool NT, bool LUT>
__global__ __launch_bounds__(PMX_BLOCK) void pmx_expand_kernel(PmxExpandParams p)
{
    constexpr int VEC = ObsVec<DT>::VEC;
    __shared__ uint32_t tab[4][8 * 32 * 32 / 32 + 8];
    __shared__ __align__(16) uint32_t lut[LUT ? (DT == 0 ? 16 * 4 : (DT == 1 ? 256 * 4 : 256 * 2)) : 4];
    if (LUT) {
        const uint32_t i = threadIdx.x;
        if (DT == 0) {
            if (i < 16) *reinterpret_cast<uint4 *>(&lut[4 * i]) = pack_obs<0>(i);
        } else if (DT == 1) {
            *reinterpret_cast<uint4 *>(&lut[4 * i]) = pack_obs<1>(i);
        } else {
            lut[2 * i] = ((i & 15u) * 0x00204081u) & 0x01010101u;
            lut[2 * i + 1] = ((i >> 4) * 0x00204081u) & 0x01010101u;
        }
        __syncthreads();
    }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int W = p.lay_W, H = p.lay_H;
    const int HW = H * W;
    const long total = (long)p.N * p.n_emit;
    long blk = p.reverse ? (long)gridDim.x - 1 - (long)blockIdx.x : (long)blockIdx.x;
    if (p.n_emit == 4 && (p.N & 127) == 0) {
        // XCD-aware order: blocks b, b+8, b+16, .. share an XCD and its L2.  Give each XCD 16 consecutive envs, i.e. the
        // envs whose SoA snapshot words share 64-byte lines, so that a line is fetched into one L2 instead of eight
        // (PMC: FETCH_SIZE per launch drops accordingly; wall time is unchanged, the kernel is store-bound).
        const long g = blk >> 7, r = blk & 127;
        blk = (g << 7) + ((r & 7) << 4) + (r >> 3);
    }
    const long q = blk * 4 + wave;
    if (q >= total) return;
    const long env = q / p.n_emit;
    const int slot = (int)(q - env * p.n_emit);
    const int agent = p.single_agent >= 0 ? p.single_agent : p.emit[slot];
    const uint32_t *S = p.snap[agent] + env;
    const size_t N = (size_t)p.N;
    uint32_t *T = tab[wave];
    const PmxLayoutDev *L = p.lay + (p.layout_idx ? p.layout_idx[env] : 0);

    // issue the snapshot loads first, their latency hides behind the table initialisation
    const uint32_t food = lane < H ? S[(size_t)lane * N] : 0u;
    uint32_t pt = 0;
    if (lane < 4) pt = S[(size_t)PMX_W_AGENT_A(H, lane) * N];
    else if (lane < 8) pt = S[(size_t)PMX_W_CAPS(H, (lane - 4) >> 1) * N];
    const uint32_t a_self = S[(size_t)PMX_W_AGENT_A(H, agent) * N];
    const uint32_t b_self = S[(size_t)PMX_W_AGENT_B(H, agent) * N];

    const int n_words = (8 * HW + 31) >> 5;
    const int wall_words = (HW + 31) >> 5;
    for (int k = lane; k < n_words + 1; k += 64) T[k] = k < wall_words ? L->wall_stream[k] : 0u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane < H) {
        stream_or_row(T, (uint32_t)((6 * H + lane) * W), food & L->hi_mask, W);    // blue food: x >= int(W/2) (capture.py:336)
        stream_or_row(T, (uint32_t)((7 * H + lane) * W), food & L->lo_mask, W);    // red food
    }
    if (lane < 4) {
        const int x = pt & 0xFF, y = (pt >> 8) & 0xFF;
        const int plane = lane == agent ? 1 : (((lane ^ agent) == 2) ? 4 : 5);      // gymPacMan.py:205-215
        const uint32_t off = (uint32_t)((plane * H + y) * W + x);
        atomicOr(&T[off >> 5], 1u << (off & 31));
    } else if (lane < 8) {
        const uint32_t cxy = (pt >> (16 * ((lane - 4) & 1))) & 0xFFFFu;
        if (cxy != 0xFFFFu) {
            const int x = cxy & 0xFF, y = cxy >> 8;
            const int plane = (2 * x > W) ? 2 : 3;                                  // halfList: blue x > W/2, red x <= W/2
            const uint32_t off = (uint32_t)((plane * H + y) * W + x);
            atomicOr(&T[off >> 5], 1u << (off & 31));
        }
    }
    const uint32_t carry = (b_self >> 8) & 0xFFF;
    const int fself = (H + (int)((a_self >> 8) & 0xFF)) * W + (int)(a_self & 0xFF);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    const int n_vec = 8 * HW / VEC;
    uint4 *out = reinterpret_cast<uint4 *>(p.obs) + (size_t)q * n_vec;
    for (int k = lane; k < n_vec; k += 64) {
        const uint32_t e0 = (uint32_t)k * VEC;
        const uint32_t bits = T[e0 >> 5] >> (e0 & 31);
        uint4 v;
        if (LUT) {
            if (DT == 0) v = *reinterpret_cast<const uint4 *>(&lut[(bits & 15u) * 4]);
            else if (DT == 1) v = *reinterpret_cast<const uint4 *>(&lut[(bits & 255u) * 4]);
            else {
                const uint2 lo = *reinterpret_cast<const uint2 *>(&lut[(bits & 255u) * 2]);
                const uint2 hi = *reinterpret_cast<const uint2 *>(&lut[((bits >> 8) & 255u) * 2]);
                v = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
        } else {
            v = pack_obs<DT>(bits);
        }
        const uint32_t d = (uint32_t)(fself - (int)e0);
        if (d < (uint32_t)VEC) patch_self<DT>(v, (int)d, carry);
        if (NT) {   // streaming stores (merged into one dwordx4 nt): used when the planes exceed the Infinity Cache
            __builtin_nontemporal_store(v.x, &out[k].x); __builtin_nontemporal_store(v.y, &out[k].y);
            __builtin_nontemporal_store(v.z, &out[k].z); __builtin_nontemporal_store(v.w, &out[k].w);
        } else {
            out[k] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The same expansion with ONE wavefront per ENV for the narrow element types: at 1 or 2 bytes per element an (env, agent)
// block is 1.2 - 2.5 KB, i.e. one or two store instructions per lane behind a fixed ~2 us of snapshot-load latency and table
// set-up -- pmx_expand_kernel is issue / latency bound there (0.39 of the HBM peak for uint8).  Here a wave issues the snapshot
// loads of all four agents together, builds the four bit streams side by side in LDS and then streams the env's whole
// [4][8][H][W] block (4.9 KB of uint8 for smallCapture, contiguous: whole 128-byte lines except at the two ends, where
// pmx_expand_kernel's 1 232-byte agent blocks split a line each).
// ---------------------------------------------------------------------------------------------------------------
// EPW envs per wave, consecutive, with the NEXT env's snapshot words loaded while the current env's planes are streamed.
// Measured and NOT used (EPW = 1 is what is launched): s_memtime (tools/rule_ticks.py, uint8) shows a wave spending ~7 k of its
// ~14 k ticks on the snapshot loads, which queue behind the other waves' plane stores, yet two envs per wave with the prefetch ran
// 20.4 us against 18.5 us -- as did a variant with 16 envs per block and cooperative, fully coalesced snapshot loads (24.9 us):
// with half the waves the chip has fewer store streams in flight, and that costs more than the hidden latency gains.
struct SnapWords { uint32_t food[4], pt[4], a_self[4], b_self[4]; };
__device__ __forceinline__ void load_snap_words(SnapWords &s, const PmxExpandParams &p, long env, int lane, int H)
{
    const size_t N = (size_t)p.N;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const uint32_t *S = p.snap[a] + env;
        s.food[a] = lane < H ? S[(size_t)lane * N] : 0u;
        s.pt[a] = 0;
        if (lane < 4) s.pt[a] = S[(size_t)PMX_W_AGENT_A(H, lane) * N];
        else if (lane < 8) s.pt[a] = S[(size_t)PMX_W_CAPS(H, (lane - 4) >> 1) * N];
        s.a_self[a] = S[(size_t)PMX_W_AGENT_A(H, a) * N];
        s.b_self[a] = S[(size_t)PMX_W_AGENT_B(H, a) * N];
    }
}

template <int DT, bool NT, int EPW>
__global__ __launch_bounds__(PMX_BLOCK) void pmx_expand4_kernel(PmxExpandParams p)
{
    constexpr int VEC = ObsVec<DT>::VEC;
    constexpr int TW = 8 * 32 * 32 / 32 + 8;
    __shared__ uint32_t tab[4][4][TW];                        // [wave][agent][stream words]
    __shared__ __align__(16) uint32_t lut[DT == 0 ? 16 * 4 : (DT == 1 ? 256 * 4 : 256 * 2)];
    {
        const uint32_t i = threadIdx.x;
        if (DT == 0) {
            if (i < 16) *reinterpret_cast<uint4 *>(&lut[4 * i]) = pack_obs<0>(i);
        } else if (DT == 1) {
            *reinterpret_cast<uint4 *>(&lut[4 * i]) = pack_obs<1>(i);
        } else {
            lut[2 * i] = ((i & 15u) * 0x00204081u) & 0x01010101u;
            lut[2 * i + 1] = ((i >> 4) * 0x00204081u) & 0x01010101u;
        }
        __syncthreads();
    }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int W = p.lay_W, H = p.lay_H, HW = H * W;
    long blk = p.reverse ? (long)gridDim.x - 1 - (long)blockIdx.x : (long)blockIdx.x;
    if ((p.N & (128 * EPW - 1)) == 0 && EPW <= 4) {
        // XCD-aware order (blocks b, b + 8, .. share an XCD): the 16 envs whose SoA snapshot words share a 64-byte line are
        // 4 / EPW consecutive blocks; send them to one XCD
        constexpr int G = 4 / EPW;                            // blocks per line of 16 envs
        const long r = blk & (8 * G - 1);
        blk = (blk & ~(long)(8 * G - 1)) + (r & 7) * G + (r >> 3);
    }
    const long env_first = (blk * 4 + wave) * EPW;
    if (env_first >= p.N) return;
#ifdef PMX_RULE_TIMING
    unsigned long long last_ = __builtin_amdgcn_s_memtime();
    const bool rec_ = (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2 || blockIdx.x == gridDim.x - 1) && threadIdx.x == 0;
#define PMX_XTICK(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); if (rec_) atomicAdd(&pmx_rule_ticks[8 + i], now_ - last_); last_ = now_; } while (0)
#else
#define PMX_XTICK(i) do { } while (0)
#endif
    const int n_words = (8 * HW + 31) >> 5;
    const int wall_words = (HW + 31) >> 5;
    const int n_vec = 8 * HW / VEC;
    SnapWords cur;
    load_snap_words(cur, p, env_first, lane, H);              // all four agents' snapshot words in flight together
#pragma unroll 1
    for (int q = 0; q < EPW; ++q) {
        const long env = env_first + q;
        if (env >= p.N) break;
        const PmxLayoutDev *L = p.lay + (p.layout_idx ? p.layout_idx[env] : 0);
#pragma unroll
        for (int a = 0; a < 4; ++a)
            for (int k = lane; k < n_words + 1; k += 64) tab[wave][a][k] = k < wall_words ? L->wall_stream[k] : 0u;
        const uint32_t hi_mask = L->hi_mask, lo_mask = L->lo_mask;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        PMX_XTICK(0);
        int fself[4];
        uint32_t carry[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            uint32_t *T = tab[wave][a];
            if (lane < H) {
                stream_or_row(T, (uint32_t)((6 * H + lane) * W), cur.food[a] & hi_mask, W);
                stream_or_row(T, (uint32_t)((7 * H + lane) * W), cur.food[a] & lo_mask, W);
            }
            if (lane < 4) {
                const int x = cur.pt[a] & 0xFF, y = (cur.pt[a] >> 8) & 0xFF;
                const int plane = lane == a ? 1 : (((lane ^ a) == 2) ? 4 : 5);
                const uint32_t off = (uint32_t)((plane * H + y) * W + x);
                atomicOr(&T[off >> 5], 1u << (off & 31));
            } else if (lane < 8) {
                const uint32_t cxy = (cur.pt[a] >> (16 * ((lane - 4) & 1))) & 0xFFFFu;
                if (cxy != 0xFFFFu) {
                    const int x = cxy & 0xFF, y = cxy >> 8;
                    const int plane = (2 * x > W) ? 2 : 3;
                    const uint32_t off = (uint32_t)((plane * H + y) * W + x);
                    atomicOr(&T[off >> 5], 1u << (off & 31));
                }
            }
            carry[a] = (cur.b_self[a] >> 8) & 0xFFF;
            fself[a] = (H + (int)((cur.a_self[a] >> 8) & 0xFF)) * W + (int)(cur.a_self[a] & 0xFF);
        }
        if (EPW > 1 && q + 1 < EPW && env + 1 < p.N) load_snap_words(cur, p, env + 1, lane, H);   // in flight behind the stores below
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        PMX_XTICK(1);

        uint4 *out = reinterpret_cast<uint4 *>(p.obs) + (size_t)env * 4 * n_vec;
        // U iterations of the store loop at a time: all their stream words are read first, then all look-up-table entries,
        // then the stores are issued -- one LDS round trip per group instead of two per 16 bytes (a wave spent 1.2 k ticks per
        // store instruction on these dependent reads)
        constexpr int U = 5;
        for (int base = 0; base < 4 * n_vec; base += 64 * U) {
            uint32_t bits[U];
            int ag[U];
            uint32_t e0s[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int kk = base + 64 * u + lane;
                const int a = (kk >= n_vec) + (kk >= 2 * n_vec) + (kk >= 3 * n_vec);
                const int a_c = kk < 4 * n_vec ? a : 0;
                const int k = kk < 4 * n_vec ? kk - a * n_vec : 0;
                ag[u] = a_c;
                e0s[u] = (uint32_t)k * VEC;
                bits[u] = tab[wave][a_c][e0s[u] >> 5] >> (e0s[u] & 31);
            }
            uint4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (DT == 0) v[u] = *reinterpret_cast<const uint4 *>(&lut[(bits[u] & 15u) * 4]);
                else if (DT == 1) v[u] = *reinterpret_cast<const uint4 *>(&lut[(bits[u] & 255u) * 4]);
                else {
                    const uint2 lo = *reinterpret_cast<const uint2 *>(&lut[(bits[u] & 255u) * 2]);
                    const uint2 hi = *reinterpret_cast<const uint2 *>(&lut[((bits[u] >> 8) & 255u) * 2]);
                    v[u] = make_uint4(lo.x, lo.y, hi.x, hi.y);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int kk = base + 64 * u + lane;
                const int a = ag[u];
                const int fs = a == 0 ? fself[0] : (a == 1 ? fself[1] : (a == 2 ? fself[2] : fself[3]));
                const uint32_t cr = a == 0 ? carry[0] : (a == 1 ? carry[1] : (a == 2 ? carry[2] : carry[3]));
                const uint32_t d = (uint32_t)(fs - (int)e0s[u]);
                if (d < (uint32_t)VEC) patch_self<DT>(v[u], (int)d, cr);
                if (kk < 4 * n_vec) {
                    if (NT) {
                        __builtin_nontemporal_store(v[u].x, &out[kk].x); __builtin_nontemporal_store(v[u].y, &out[kk].y);
                        __builtin_nontemporal_store(v[u].z, &out[kk].z); __builtin_nontemporal_store(v[u].w, &out[kk].w);
                    } else {
                        out[kk] = v[u];
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // the tables are rebuilt for the next env
        __builtin_amdgcn_wave_barrier();
        PMX_XTICK(2);
    }
#ifdef PMX_RULE_TIMING
    if (rec_) atomicAdd(&pmx_rule_ticks[14], 1ull);
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// Training-side emission: what the MAPPO rollout does with every tick's observations (pacman_mappo_resnet.py:462-469),
// straight from the snapshots: the two learners' planes -- canonicalised for a red team (canonicalize_obs :215-229: x
// flipped, capsule planes 2 <-> 3 and food planes 6 <-> 7 swapped) -- and merge_obs_for_critic (:267-274: the first
// learner's planes with plane 4 cleared and plane 1 = max of both learners' self planes).  Slots 0, 1 = the learners,
// slot 2 = the merged input; same packed-bit-stream construction and look-up-table expansion as pmx_expand_kernel, with
// the flip applied while the stream is built (rows bit-reversed, x -> W-1-x).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t flip_row(uint32_t v, int W, bool flip) { return flip ? (__builtin_bitreverse32(v) >> (32 - W)) : v; }

// One wavefront per ENV.  The two learners' streams are assembled side by side, one per half of the wave (lanes 0-31: the
// first learner's snapshot, lanes 32-63: the second's; H <= 32 rows each), and the merged slot is the first learner's
// stream patched in registers while it is expanded (the ally bit of plane 4 cleared, the mate's bit of plane 1 set), so no
// third stream is built.  The 16-byte vectors of all slots are then dealt to the lanes as ONE index range (2 or 3 x n_vec):
// on smallCapture 231 vectors fill four store instructions to 90 % where one wave per (env, slot) -- the round-2 kernel --
// filled six to 60 %, and the set-up (snapshot loads, stream assembly, the block's look-up table) is paid once per env
// instead of once per slot: 28.3 -> 14.5 us at 16 384 smallCapture envs, 12.3 -> 10.0 / 20.3 -> 17.0 us on the 20 x 20 boards
// (4 096 / 8 192 envs).  U = store instructions per group of look-ups (1: 16.0 us, 2: 14.5, 4: 14.4, 5: 15.0 on smallCapture).
template <int DT>
__global__ __launch_bounds__(PMX_BLOCK) void pmx_emit_team_kernel(PmxEmitParams p)
{
    constexpr int VEC = ObsVec<DT>::VEC;
    constexpr int U = 2;
    constexpr int TW = 8 * 32 * 32 / 32 + 8;
    __shared__ uint32_t tab[4][2][TW];                        // [wave][learner][stream words]
    __shared__ __align__(16) uint32_t lut[DT == 0 ? 16 * 4 : (DT == 1 ? 256 * 4 : 256 * 2)];
    {
        const uint32_t i = threadIdx.x;
        if (DT == 0) {
            if (i < 16) *reinterpret_cast<uint4 *>(&lut[4 * i]) = pack_obs<0>(i);
        } else if (DT == 1) {
            *reinterpret_cast<uint4 *>(&lut[4 * i]) = pack_obs<1>(i);
        } else {
            lut[2 * i] = ((i & 15u) * 0x00204081u) & 0x01010101u;
            lut[2 * i + 1] = ((i >> 4) * 0x00204081u) & 0x01010101u;
        }
        __syncthreads();
    }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int half = lane >> 5, hl = lane & 31;
    const int W = p.lay_W, H = p.lay_H, HW = H * W;
    long blk = (long)blockIdx.x;
    if ((p.N & 127) == 0) {                                   // the 16 envs of one 64-byte line of snapshot words: one XCD
        const long r = blk & 31;
        blk = (blk & ~31L) + (r & 7) * 4 + (r >> 3);
    }
    const long env = blk * 4 + wave;
    if (env >= p.N) return;
    const bool flip = p.red != 0;
    const int first = p.red ? 0 : 1, second = first + 2;
    const int agent = half ? second : first;
    const size_t N = (size_t)p.N;
    const uint32_t *S = (half ? p.snap[second] : p.snap[first]) + env;
    uint32_t *T = tab[wave][half];
    const PmxLayoutDev *L = p.lay + (p.layout_idx ? p.layout_idx[env] : 0);

    const uint32_t food = hl < H ? S[(size_t)hl * N] : 0u;
    uint32_t pt = 0;
    if (hl < 4) pt = S[(size_t)PMX_W_AGENT_A(H, hl) * N];
    else if (hl < 8) pt = S[(size_t)PMX_W_CAPS(H, (hl - 4) >> 1) * N];
    const uint32_t a_self = S[(size_t)PMX_W_AGENT_A(H, agent) * N];
    const uint32_t b_self = S[(size_t)PMX_W_AGENT_B(H, agent) * N];
    const int n_words = (8 * HW + 31) >> 5;
    const int wall_words = (HW + 31) >> 5;
    for (int k = hl; k < n_words + 1; k += 32) T[k] = (!flip && k < wall_words) ? L->wall_stream[k] : 0u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (hl < H) {
        if (flip) stream_or_row(T, (uint32_t)(hl * W), flip_row(L->walls[hl], W, true), W);
        const uint32_t blue = flip_row(food & L->hi_mask, W, flip), red = flip_row(food & L->lo_mask, W, flip);
        stream_or_row(T, (uint32_t)(((flip ? 7 : 6) * H + hl) * W), blue, W);
        stream_or_row(T, (uint32_t)(((flip ? 6 : 7) * H + hl) * W), red, W);
    }
    uint32_t pt_off = 0;                                      // lanes 0-3 of each half: that agent's bit offset in the stream
    if (hl < 4) {
        const int x0 = pt & 0xFF, y = (pt >> 8) & 0xFF, x = flip ? W - 1 - x0 : x0;
        const int plane = hl == agent ? 1 : (((hl ^ agent) == 2) ? 4 : 5);
        pt_off = (uint32_t)((plane * H + y) * W + x);
        atomicOr(&T[pt_off >> 5], 1u << (pt_off & 31));
    } else if (hl < 8) {
        const uint32_t cxy = (pt >> (16 * ((hl - 4) & 1))) & 0xFFFFu;
        if (cxy != 0xFFFFu) {
            const int x0 = cxy & 0xFF, y = cxy >> 8, x = flip ? W - 1 - x0 : x0;
            const int plane0 = (2 * x0 > W) ? 2 : 3;
            const int plane = flip ? 5 - plane0 : plane0;
            const uint32_t off = (uint32_t)((plane * H + y) * W + x);
            atomicOr(&T[off >> 5], 1u << (off & 31));
        }
    }
    const int sx = (int)(a_self & 0xFF);
    const int fself_v = (H + (int)((a_self >> 8) & 0xFF)) * W + (flip ? W - 1 - sx : sx);
    const int fself0 = __builtin_amdgcn_readlane(fself_v, 0), fself1 = __builtin_amdgcn_readlane(fself_v, 32);
    const uint32_t carry0 = (__builtin_amdgcn_readlane((int)b_self, 0) >> 8) & 0xFFF;
    const uint32_t carry1 = (__builtin_amdgcn_readlane((int)b_self, 32) >> 8) & 0xFFF;
    // merged slot (merge_obs_for_critic): plane 4 of the first learner's stream cleared, the second learner -- where ITS OWN
    // snapshot has it -- added to plane 1; both learners on one cell: max(1 + carry, 1 + carry')
    const int f_ally = second == 2 ? __builtin_amdgcn_readlane((int)pt_off, 2) : __builtin_amdgcn_readlane((int)pt_off, 3);
    int fmate = fself1;
    uint32_t carry_m = carry0;
    if (fmate == fself0) {
        carry_m = carry0 > carry1 ? carry0 : carry1;
        fmate = -1;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    const int n_vec = 8 * HW / VEC;
    const int total = (p.merged ? 3 : 2) * n_vec;
    uint4 *out_team = reinterpret_cast<uint4 *>(p.team_obs) + (size_t)env * 2 * n_vec;            // slots 0, 1 are contiguous
    uint4 *out_merged = reinterpret_cast<uint4 *>(p.merged) + (size_t)env * n_vec - 2 * n_vec;   // indexed by kk >= 2 n_vec
    const uint32_t *T0 = tab[wave][0], *T1 = tab[wave][1];
    for (int base = 0; base < total; base += 64 * U) {
        uint32_t bits[U], e0s[U];
        int sl[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kk = base + 64 * u + lane;
            const int s = (kk >= n_vec) + (kk >= 2 * n_vec);
            const bool ok = kk < total;
            sl[u] = ok ? s : 0;
            e0s[u] = ok ? (uint32_t)(kk - s * n_vec) * VEC : 0u;
            bits[u] = (sl[u] == 1 ? T1 : T0)[e0s[u] >> 5] >> (e0s[u] & 31);
            if (sl[u] == 2) {
                const uint32_t d4 = (uint32_t)(f_ally - (int)e0s[u]);
                if (d4 < (uint32_t)VEC) bits[u] &= ~(1u << d4);
                const uint32_t d2 = (uint32_t)(fmate - (int)e0s[u]);
                if (fmate >= 0 && d2 < (uint32_t)VEC) bits[u] |= 1u << d2;
            }
        }
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (DT == 0) v[u] = *reinterpret_cast<const uint4 *>(&lut[(bits[u] & 15u) * 4]);
            else if (DT == 1) v[u] = *reinterpret_cast<const uint4 *>(&lut[(bits[u] & 255u) * 4]);
            else {
                const uint2 lo = *reinterpret_cast<const uint2 *>(&lut[(bits[u] & 255u) * 2]);
                const uint2 hi = *reinterpret_cast<const uint2 *>(&lut[((bits[u] >> 8) & 255u) * 2]);
                v[u] = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kk = base + 64 * u + lane;
            const int fs = sl[u] == 1 ? fself1 : fself0;
            const uint32_t cr = sl[u] == 0 ? carry0 : (sl[u] == 1 ? carry1 : carry_m);
            const uint32_t d = (uint32_t)(fs - (int)e0s[u]);
            if (d < (uint32_t)VEC) patch_self<DT>(v[u], (int)d, cr);
            if (sl[u] == 2 && fmate >= 0) {
                const uint32_t d2 = (uint32_t)(fmate - (int)e0s[u]);
                if (d2 < (uint32_t)VEC) patch_self<DT>(v[u], (int)d2, carry1);
            }
            if (kk < total) (sl[u] == 2 ? out_merged : out_team)[kk] = v[u];
        }
    }
}

// ev0 / ev1 (both or neither): start / stop events of this very dispatch, as in pmx_launch_expand
extern "C" hipError_t pmx_launch_emit_team(const PmxEmitParams *p, int dtype, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1)
{
    const unsigned blocks = (unsigned)(((long)p->N + 3) / 4);    // one wave per env
#define PMX_EMIT(DT)                                                                                                          \
    do {                                                                                                                      \
        if (ev0) hipExtLaunchKernelGGL(pmx_emit_team_kernel<DT>, dim3(blocks), dim3(PMX_BLOCK), 0, st, ev0, ev1, 0, *p);       \
        else hipLaunchKernelGGL(pmx_emit_team_kernel<DT>, dim3(blocks), dim3(PMX_BLOCK), 0, st, *p);                           \
    } while (0)
    switch (dtype) {
    case 0: PMX_EMIT(0); break;
    case 1: PMX_EMIT(1); break;
    default: PMX_EMIT(2); break;
    }
#undef PMX_EMIT
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// host-side launchers (called by the C ABI in pmx_api.hip)
// ---------------------------------------------------------------------------------------------------------------
// ev0/ev1 (both or neither): events that receive the START and STOP timestamps of this very dispatch (hipExtLaunchKernelGGL),
// i.e. the kernel's own duration as a profiler sees it, without the dispatch gaps a hipEventRecord pair would include
extern "C" hipError_t pmx_launch_rule(const PmxTickParams *p, int H, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1)
{
    const int blocks = (p->N + PMX_RULE_BLOCK - 1) / PMX_RULE_BLOCK;
    const size_t lds = (p->layout_idx ? 32 + (size_t)(3 * PMX_MAX_H_LDS + 32 + 16) * PMX_RULE_BLOCK : 32 + (size_t)(3 * H + 16) * PMX_RULE_BLOCK) * sizeof(uint32_t);
    // row loops unrolled for the board-height bucket (see load_env_issue)
    const int hb = H <= 12 ? 12 : (H <= 16 ? 16 : (H <= 20 ? 20 : 32));
#define PMX_RULE_LAUNCH(B, HBV)                                                                                         \
    do {                                                                                                                \
        if (ev0) hipExtLaunchKernelGGL((pmx_rule_kernel<B, HBV>), dim3(blocks), dim3(PMX_RULE_BLOCK), (uint32_t)lds, st, ev0, ev1, 0, *p); \
        else hipLaunchKernelGGL((pmx_rule_kernel<B, HBV>), dim3(blocks), dim3(PMX_RULE_BLOCK), lds, st, *p);              \
    } while (0)
#define PMX_RULE_PICK(B)                                                                      \
    switch (hb) {                                                                             \
    case 12: PMX_RULE_LAUNCH(B, 12); break;                                                   \
    case 16: PMX_RULE_LAUNCH(B, 16); break;                                                   \
    case 20: PMX_RULE_LAUNCH(B, 20); break;                                                   \
    default: PMX_RULE_LAUNCH(B, 32); break;                                                   \
    }
    if (p->dist) { PMX_RULE_PICK(true) } else { PMX_RULE_PICK(false) }
    return hipGetLastError();
}

extern "C" hipError_t pmx_launch_rule_agent(const PmxTickParams *p, int H, int agent, hipStream_t st)
{
    const int blocks = (p->N + PMX_RULE_BLOCK - 1) / PMX_RULE_BLOCK;
    const size_t lds = (p->layout_idx ? 32 + (size_t)(3 * PMX_MAX_H_LDS + 32 + 16) * PMX_RULE_BLOCK : 32 + (size_t)(3 * H + 16) * PMX_RULE_BLOCK) * sizeof(uint32_t);
    if (p->dist) hipLaunchKernelGGL(pmx_rule_agent_kernel<true>, dim3(blocks), dim3(PMX_RULE_BLOCK), lds, st, *p, agent);
    else hipLaunchKernelGGL(pmx_rule_agent_kernel<false>, dim3(blocks), dim3(PMX_RULE_BLOCK), lds, st, *p, agent);
    return hipGetLastError();
}

extern "C" hipError_t pmx_launch_successor(const PmxTickParams *p, int H, int agent, hipStream_t st)
{
    const int blocks = (p->N + PMX_RULE_BLOCK - 1) / PMX_RULE_BLOCK;
    const size_t lds = (p->layout_idx ? 32 + (size_t)(3 * PMX_MAX_H_LDS + 32 + 16) * PMX_RULE_BLOCK : 32 + (size_t)(3 * H + 16) * PMX_RULE_BLOCK) * sizeof(uint32_t);
    hipLaunchKernelGGL(pmx_successor_kernel, dim3(blocks), dim3(PMX_RULE_BLOCK), lds, st, *p, agent);
    return hipGetLastError();
}

extern "C" hipError_t pmx_launch_reset(const PmxTickParams *p, int H, hipStream_t st)
{
    const int blocks = (p->N + PMX_RULE_BLOCK - 1) / PMX_RULE_BLOCK;
    const size_t lds = (p->layout_idx ? 32 + (size_t)(3 * PMX_MAX_H_LDS + 32 + 16) * PMX_RULE_BLOCK : 32 + (size_t)(3 * H + 16) * PMX_RULE_BLOCK) * sizeof(uint32_t);
    hipLaunchKernelGGL(pmx_reset_kernel, dim3(blocks), dim3(PMX_RULE_BLOCK), lds, st, *p);
    return hipGetLastError();
}

extern "C" hipError_t pmx_launch_expand(const PmxExpandParams *p, const PmxExpandTuning *tune, int dtype, hipStream_t st, hipEvent_t ev0,
                                         hipEvent_t ev1)
{
    const long waves = (long)p->N * p->n_emit;
    const unsigned blocks = (unsigned)((waves + 3) / 4);
    // Store policy, measured in one process with tools/ab_expand.py (us per launch):
    //   f32   323 MB  ordinary 53.7 | nt 64.9 | nt + 3 blocks/CU 60.0
    //         646 MB  ordinary 114.6 | nt 126.1 | nt + 3 blocks/CU 107.2
    //        1.29 GB  ordinary 262 | nt 246 | nt + 4 blocks/CU 227 | nt + 3 blocks/CU 207.8 | nt + 2 blocks/CU 278
    //   bf16  161 MB  31.6/35.9   323 MB 61.1/66.7  646 MB 172/124 (ordinary / nt)
    //   u8     80 MB  29.6/28.5   323 MB 111/102
    // Ordinary stores win while the 256 MiB Infinity Cache can absorb a good part of the planes.  Beyond that, streaming
    // (non-temporal) stores win, and they win more with FEWER wavefronts in flight: every wave is its own sequential
    // write stream, and 12 waves per CU instead of 32 keep the number of DRAM pages open at once low enough for the
    // write-combined bursts to stay row-buffer hits (6.2 TB/s instead of 5.3 at 1.29 GB).  The occupancy cap is an
    // unused dynamic-LDS reservation of 40 000 bytes per block (3 blocks of 4 waves fit the 160 KB of a CU).
    // (the cap is for float32 only: the bf16 / uint8 variants are issue-bound and lose up to 2x with it)
    //
    // Alternating sweep: a caller steps the same observation buffer tick after tick, and the last part written in tick t is
    // still in the Infinity Cache when tick t+1 starts.  Walking the blocks in the opposite direction every other tick makes
    // tick t+1 overwrite exactly those lines first, while they are still cached, so they never cost an HBM write:
    //   f32 small 323 MB  51.4 -> 43.8 us      blox 419 MB  75.7 -> 60.4      blox 839 MB  151.8 -> 125.1
    //       small 646 MB 118.7 -> 108.6 (streaming + cap: 105-107)       small 1.29 GB  271 -> 257 (streaming + cap: 208)
    // It needs ordinary (cache-allocating) stores, so float32 planes up to 900 MB now use them; p->reverse carries the parity.
    const size_t elem = dtype == 0 ? 4 : (dtype == 1 ? 2 : 1);
    const size_t bytes = (size_t)waves * 8 * p->lay_H * p->lay_W * elem;
    bool nt = elem == 1 || (elem == 2 && bytes > ((size_t)512 << 20)) || (elem == 4 && bytes > ((size_t)900 << 20));
    if (tune && tune->nt >= 0) nt = tune->nt != 0;                        // experiment override (pmx_set_tuning)
    size_t lds_pad = (nt && elem == 4) ? 40000 : 0;
    if (tune && tune->lds_pad >= 0) lds_pad = (size_t)tune->lds_pad;      // experiment override
    PmxExpandParams q = *p;
    if (nt) q.reverse = 0;
    p = &q;
    bool use_lut = true;
    if (tune && tune->lut >= 0) use_lut = tune->lut != 0;                 // experiment override
    // uint8 planes with all four agents emitted: one wave per env (pmx_expand4_kernel; measured at 16 384 envs of smallCapture:
    // 25.8 -> 22.4 us for uint8, but 26.7 -> 28.3 us for bfloat16, which therefore keeps the wave per (env, agent));
    // pmx_set_tuning("expand_wave_per_env", 0/1) overrides
    bool per_env = dtype == 2 && p->n_emit == 4 && p->single_agent < 0;
    if (tune && tune->per_env >= 0) per_env = tune->per_env != 0 && p->n_emit == 4 && p->single_agent < 0;
    if (per_env) {
        constexpr int epw = 1;                                 // envs per wave (see pmx_expand4_kernel)
        const unsigned b4 = (unsigned)((p->N + 4 * epw - 1) / (4 * epw));
#define PMX_EXPAND4(DT, NTV)                                                                                              \
    do {                                                                                                                  \
        if (ev0) hipExtLaunchKernelGGL((pmx_expand4_kernel<DT, NTV, epw>), dim3(b4), dim3(PMX_BLOCK), 0, st, ev0, ev1, 0, *p); \
        else hipLaunchKernelGGL((pmx_expand4_kernel<DT, NTV, epw>), dim3(b4), dim3(PMX_BLOCK), 0, st, *p);                 \
    } while (0)
        // the planes of these types fit the Infinity Cache up to a few hundred MB: ordinary stores there, streaming beyond
        const bool nt4 = (tune && tune->nt >= 0) ? tune->nt != 0 : bytes > ((size_t)200 << 20);
        PmxExpandParams q4 = *p;
        if (nt4) q4.reverse = 0;
        p = &q4;
        switch (dtype) {
        case 0: if (nt4) PMX_EXPAND4(0, true); else PMX_EXPAND4(0, false); break;
        case 1: if (nt4) PMX_EXPAND4(1, true); else PMX_EXPAND4(1, false); break;
        default: if (nt4) PMX_EXPAND4(2, true); else PMX_EXPAND4(2, false); break;
        }
#undef PMX_EXPAND4
        return hipGetLastError();
    }
#define PMX_EXPAND_LAUNCH2(DT, NTV, LUTV)                                                                               \
    do {                                                                                                                \
        if (ev0) hipExtLaunchKernelGGL((pmx_expand_kernel<DT, NTV, LUTV>), dim3(blocks), dim3(PMX_BLOCK), (uint32_t)lds_pad, st, ev0, ev1, 0, *p); \
        else hipLaunchKernelGGL((pmx_expand_kernel<DT, NTV, LUTV>), dim3(blocks), dim3(PMX_BLOCK), lds_pad, st, *p);      \
    } while (0)
#define PMX_EXPAND_LAUNCH1(DT, NTV)                                                                                     \
    do {                                                                                                                \
        if (use_lut) PMX_EXPAND_LAUNCH2(DT, NTV, true); else PMX_EXPAND_LAUNCH2(DT, NTV, false);                        \
    } while (0)
#define PMX_EXPAND_LAUNCH(DT)                                                                                           \
    do {                                                                                                                \
        if (nt) PMX_EXPAND_LAUNCH1(DT, true); else PMX_EXPAND_LAUNCH1(DT, false);                                       \
    } while (0)
    switch (dtype) {
    case 0: PMX_EXPAND_LAUNCH(0); break;
    case 1: PMX_EXPAND_LAUNCH(1); break;
    default: PMX_EXPAND_LAUNCH(2); break;
    }
    return hipGetLastError();
}

#ifdef PMX_RULE_TIMING
extern "C" int pmx_rule_ticks_read(unsigned long long *out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(pmx_rule_ticks), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(pmx_rule_ticks), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
