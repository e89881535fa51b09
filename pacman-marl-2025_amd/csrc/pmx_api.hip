// pmx_api.hip -- the C ABI of include/pmx.h on top of the HIP kernels (gfx950).
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/pmx.h"
#include "pmx_device.h"

extern "C" hipError_t pmx_launch_rule(const PmxTickParams *p, int H, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1);
extern "C" hipError_t pmx_launch_rule_agent(const PmxTickParams *p, int H, int agent, hipStream_t st);
extern "C" hipError_t pmx_launch_reset(const PmxTickParams *p, int H, hipStream_t st);
extern "C" hipError_t pmx_launch_successor(const PmxTickParams *p, int H, int agent, hipStream_t st);
extern "C" hipError_t pmx_launch_emit_team(const PmxEmitParams *p, int dtype, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1);
extern "C" hipError_t pmx_launch_expand(const PmxExpandParams *p, const PmxExpandTuning *tune, int dtype, hipStream_t st, hipEvent_t ev0,
                                         hipEvent_t ev1);
extern "C" hipError_t pmx_launch_maze(const PmxLayoutDev *lay_dev, const int16_t *cell_index_dev, int n_cells,
                                      const int8_t *cells_dev, uint8_t *dist_dev, hipStream_t st);

struct pmx_layout_host {
    PmxLayoutDev dev;
    std::vector<int8_t> cells;       // open cells in asList(False) order
    std::vector<int16_t> cell_index;
};

struct pmx_env {
    pmx_config cfg;
    PmxLayoutDev lay;          // host copy of layout 0 (dimensions are common to all layouts)
    std::vector<pmx_layout_host> layouts;
    std::vector<int32_t> layout_index;   // per env (empty: one layout)
    int32_t *layout_idx_dev;
    uint8_t *dist_dev;           // maze-distance matrices of every layout, back to back (bots); NULL if too many layouts
    int16_t *cell_index_dev;     // [n_layouts][1024]
    PmxLayoutDev *lay_dev;
    int8_t *dump_dev;
    uint32_t *state_dev;
    uint32_t *snap_dev;
    int n_emit;
    int emit[4];
    int elem_bytes;
    int open_agent;            // next agent expected by pmx_step_agent
    // optional per-kernel timing (pmx_profile_begin/end): pairs of events around each launch
    bool profiling;
    uint64_t expand_launches = 0;   // parity selects the direction of the expansion sweep
    PmxExpandTuning tune;           // launch tuning, read from the environment once at pmx_create
    bool snaps_valid = false;       // the three sub-step snapshots belong to the current state (set by pmx_step)
    std::vector<hipEvent_t> ev_rule, ev_expand;
    size_t ev_rule_used, ev_expand_used;
};

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(PMX_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));    \
    } while (0)

// Order in which the FIFO BFS of dumpFoodFromDeath (capture.py:635-661) first visits offsets from the death
// cell: start (0,0); pop; skip if seen; expand the 3x3 neighbourhood with dx outer, dy inner.
std::vector<int8_t> dump_order(int R)
{
    const int span = 2 * R + 1;
    std::vector<uint8_t> seen((size_t)span * span, 0);
    std::vector<std::pair<int, int>> q;
    q.reserve((size_t)span * span * 9 + 1);
    q.emplace_back(0, 0);
    std::vector<int8_t> out;
    for (size_t head = 0; head < q.size(); ++head) {
        auto [x, y] = q[head];
        uint8_t &s = seen[(size_t)(x + R) * span + (y + R)];
        if (s) continue;
        s = 1;
        out.push_back((int8_t)x);
        out.push_back((int8_t)y);
        for (int dx = -1; dx <= 1; ++dx)
            for (int dy = -1; dy <= 1; ++dy) {
                int nx = x + dx, ny = y + dy;
                if (std::abs(nx) > R || std::abs(ny) > R) continue;
                q.emplace_back(nx, ny);
            }
    }
    return out;
}

hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

void fill_tick_params(pmx_env *env, PmxTickParams &p, const int8_t *actions, const pmx_step_out *out)
{
    std::memset(&p, 0, sizeof(p));
    p.state = env->state_dev;
    p.snap = env->snap_dev;
    p.lay = env->lay_dev;
    p.layout_idx = env->layout_idx_dev;
    p.layout_idx_rw = env->cfg.redraw_layouts ? env->layout_idx_dev : nullptr;
    p.n_layouts = (int32_t)env->layouts.size();
    p.dist = env->dist_dev;
    p.cell_index = env->cell_index_dev;
    p.dump = env->dump_dev;
    p.actions = actions;
    p.N = env->cfg.n_envs;
    p.length = env->cfg.length;
    p.legal_reward = env->cfg.legal_reward;
    p.defence_reward = env->cfg.defence_reward;
    p.auto_reset = env->cfg.auto_reset;
    if (out) {
        p.reward = out->reward_dev;
        p.done = out->done_dev;
        p.legal = out->legal_dev;
        p.score_change = out->score_change_dev;
        p.score = out->score_dev;
        p.agent_out = out->agent_dev;
    }
    p.seed = env->cfg.seed;
    p.lay_W = env->lay.W; p.lay_H = env->lay.H; p.lay_half = env->lay.half; p.lay_n_dump = env->lay.n_dump;
    p.lo_mask = env->lay.lo_mask; p.hi_mask = env->lay.hi_mask;
}

// when profiling, returns the pair of events to record around the next launch of that kernel (or nullptr)
hipEvent_t *prof_pair(pmx_env *env, bool expand)
{
    if (!env->profiling) return nullptr;
    std::vector<hipEvent_t> &v = expand ? env->ev_expand : env->ev_rule;
    size_t &used = expand ? env->ev_expand_used : env->ev_rule_used;
    if (used + 2 > v.size()) return nullptr;
    hipEvent_t *p = &v[used];
    used += 2;
    return p;
}

int launch_expand(pmx_env *env, void *obs, bool from_snapshots, int single_agent, hipStream_t st)
{
    PmxExpandParams x;
    std::memset(&x, 0, sizeof(x));
    const size_t snap_sz = (size_t)PMX_SNAP_WORDS(env->lay.H) * env->cfg.n_envs;
    for (int a = 0; a < 4; ++a)
        x.snap[a] = (from_snapshots && a < 3) ? env->snap_dev + a * snap_sz : env->state_dev;
    x.lay = env->lay_dev;
    x.layout_idx = env->layout_idx_dev;
    x.obs = obs;
    x.N = env->cfg.n_envs;
    x.lay_H = env->lay.H; x.lay_W = env->lay.W;
    x.single_agent = single_agent;
    if (single_agent >= 0) {
        x.n_emit = 1;
        x.emit[0] = single_agent;
    } else {
        x.n_emit = env->n_emit;
        for (int i = 0; i < 4; ++i) x.emit[i] = env->emit[i];
    }
    if (single_agent < 0) {
        // alternate the direction of the sweep over the planes from tick to tick (see pmx_launch_expand);
        // pmx_set_tuning(env, "expand_alt", 0) switches it off (the all-bytes-to-HBM measurement of bench.py)
        if (env->tune.alt != 0) x.reverse = (int32_t)(env->expand_launches++ & 1);
    }
    hipEvent_t *ev = prof_pair(env, true);       // profiling: the dispatch's own start / stop timestamps
    HIP_TRY(pmx_launch_expand(&x, &env->tune, env->cfg.obs_dtype, st, ev ? ev[0] : nullptr, ev ? ev[1] : nullptr));
    return PMX_OK;
}

}  // namespace

extern "C" {

int pmx_version(void) { return PMX_VERSION; }
const char *pmx_last_error(void) { return g_err; }

// layout.py:95-130 output -> validated device record
static int build_layout(const pmx_config *cfg, int li, pmx_layout_host &out)
{
    const int W = cfg->width, H = cfg->height;
    const uint32_t *wall_rows = cfg->wall_rows + (size_t)li * H, *food_rows = cfg->food_rows + (size_t)li * H;
    const uint32_t *cap_rows = cfg->cap_rows + (size_t)li * H;
    const int8_t *starts = cfg->starts + (size_t)li * 8;
    const uint32_t full = W == 32 ? 0xFFFFFFFFu : ((1u << W) - 1u);
    PmxLayoutDev &L = out.dev;
    std::memset(&L, 0, sizeof(L));
    L.W = W; L.H = H; L.half = W / 2;
    L.lo_mask = (1u << L.half) - 1u;
    L.hi_mask = full & ~L.lo_mask;
    int n_caps = 0;
    uint16_t capslots[4] = { 0xFFFF, 0xFFFF, 0xFFFF, 0xFFFF };
    for (int y = 0; y < H; ++y) {
        const uint32_t w = wall_rows[y], f = food_rows[y], c = cap_rows[y];
        if ((w | f | c) & ~full) return fail(PMX_ERR_INVALID, "layout %d row %d has bits beyond the width", li, y);
        if ((w & f) || (w & c) || (f & c)) return fail(PMX_ERR_INVALID, "layout %d row %d: wall/food/capsule overlap", li, y);
        const bool border = (y == 0 || y == H - 1);
        if (border ? (w != full) : (!(w & 1u) || !((w >> (W - 1)) & 1u)))
            return fail(PMX_ERR_INVALID, "layout %d row %d: the layout must be enclosed by walls (game.py:335-350 indexes neighbours)", li, y);
        L.walls[y] = w;
        L.food0[y] = f;
        L.total_food += __builtin_popcount(f);
        for (int x = 0; x < W; ++x)
            if ((c >> x) & 1u) {
                if (n_caps == PMX_MAX_CAPSULES) return fail(PMX_ERR_UNSUPPORTED, "layout %d: more than %d capsules", li, PMX_MAX_CAPSULES);
                capslots[n_caps++] = (uint16_t)(x | (y << 8));
            }
    }
    L.capw0[0] = capslots[0] | ((uint32_t)capslots[1] << 16);
    L.capw0[1] = capslots[2] | ((uint32_t)capslots[3] << 16);
    for (int i = 0; i < 4; ++i) {
        const int sx = starts[2 * i], sy = starts[2 * i + 1];
        if (sx <= 0 || sy <= 0 || sx >= W - 1 || sy >= H - 1 || ((L.walls[sy] >> sx) & 1u))
            return fail(PMX_ERR_INVALID, "layout %d: agent %d start (%d,%d) is not an open interior cell", li, i, sx, sy);
        const bool red = 2 * sx < W;   // capture.py:325-330
        if (red != ((i & 1) == 0))
            return fail(PMX_ERR_UNSUPPORTED, "layout %d: agent %d starts on the %s half: gymPacMan.py:150,185,210 hard-codes red = agents 0,2",
                        li, i, red ? "red" : "blue");
        L.startx[i] = sx; L.starty[i] = sy;
    }
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
            if ((L.walls[y] >> x) & 1u) L.wall_stream[(y * W + x) >> 5] |= 1u << ((y * W + x) & 31);
    out.cell_index.assign(32 * 32, -1);
    out.cells.clear();
    for (int x = 0; x < W; ++x)           // Grid.asList(False): x outer, y inner (game.py:225-230)
        for (int y = 0; y < H; ++y)
            if (!((L.walls[y] >> x) & 1u)) {
                out.cell_index[y * 32 + x] = (int16_t)(out.cells.size() / 2);
                out.cells.push_back((int8_t)x);
                out.cells.push_back((int8_t)y);
            }
    return PMX_OK;
}

int pmx_create(const pmx_config *cfg, pmx_env **out)
{
    if (!cfg || !out) return fail(PMX_ERR_INVALID, "pmx_create: null argument");
    *out = nullptr;
    const int W = cfg->width, H = cfg->height;
    if (W < 8 || W > PMX_MAX_DIM || H < 3 || H > PMX_MAX_DIM)
        return fail(PMX_ERR_UNSUPPORTED, "layout %dx%d outside the supported 8..32 x 3..32", W, H);
    if (!cfg->wall_rows || !cfg->food_rows || !cfg->cap_rows || !cfg->starts)
        return fail(PMX_ERR_INVALID, "pmx_create: layout arrays missing");
    if (cfg->n_envs < 1) return fail(PMX_ERR_INVALID, "n_envs must be >= 1");
    if (cfg->obs_dtype < PMX_OBS_F32 || cfg->obs_dtype > PMX_OBS_U8) return fail(PMX_ERR_INVALID, "bad obs_dtype");
    if (cfg->obs_dtype == PMX_OBS_U8 && ((H * W) & 1))
        return fail(PMX_ERR_UNSUPPORTED, "uint8 observations need an even number of cells (16-byte rows of the stream)");
    const int n_layouts = cfg->n_layouts > 1 ? cfg->n_layouts : 1;
    if (n_layouts > 1 && !cfg->layout_index) return fail(PMX_ERR_INVALID, "n_layouts > 1 needs layout_index");
    if (cfg->redraw_layouts && n_layouts < 2) return fail(PMX_ERR_INVALID, "redraw_layouts needs a pool of n_layouts > 1");

    pmx_env *env = new (std::nothrow) pmx_env();
    if (!env) return fail(PMX_ERR_NOMEM, "host allocation failed");
    env->layouts.resize(n_layouts);
    for (int li = 0; li < n_layouts; ++li) {
        int rc = build_layout(cfg, li, env->layouts[li]);
        if (rc != PMX_OK) { delete env; return rc; }
    }
    std::vector<int8_t> dump = dump_order(std::max(W, H));
    const size_t n_dump_real = dump.size() / 2;
    while ((dump.size() / 2) % 4 != 0 || dump.size() / 2 < n_dump_real + 4) dump.push_back(127);   // the kernel reads groups of four: out-of-board filler
    size_t dist_bytes = 0;
    for (auto &l : env->layouts) {
        l.dev.n_dump = (int)n_dump_real;
        l.dev.n_cells = (int)(l.cells.size() / 2);
        l.dev.dist_off = (uint32_t)dist_bytes;
        dist_bytes += (size_t)l.dev.n_cells * l.dev.n_cells;
    }
    // the in-kernel reflex bots need every layout's distance matrix resident; only handles that ask for them pay
    if (cfg->enable_bots && dist_bytes > ((size_t)1 << 31)) { delete env; return fail(PMX_ERR_UNSUPPORTED, "enable_bots: the layouts' distance matrices exceed 2 GiB"); }
    const bool with_bots = cfg->enable_bots != 0;
    env->cfg = *cfg;
    {   // experiment overrides of the expansion launch, read HERE once, not on the tick path
        auto env_int = [](const char *name) { const char *o = getenv(name); return o ? atoi(o) : -1; };
        env->tune.alt = env_int("PMX_EXPAND_ALT");
        env->tune.nt = env_int("PMX_EXPAND_NT");
        env->tune.lds_pad = env_int("PMX_EXPAND_LDS_PAD");
        env->tune.lut = env_int("PMX_EXPAND_LUT");
        env->tune.per_env = env_int("PMX_EXPAND_PER_ENV");
    }
    env->cfg.wall_rows = env->cfg.food_rows = env->cfg.cap_rows = nullptr;
    env->cfg.starts = nullptr;
    env->cfg.layout_index = nullptr;
    env->cfg.n_layouts = n_layouts;
    env->lay = env->layouts[0].dev;
    env->open_agent = 0;
    if (n_layouts > 1) {
        env->layout_index.assign(cfg->layout_index, cfg->layout_index + cfg->n_envs);
        for (int32_t v : env->layout_index)
            if (v < 0 || v >= n_layouts) { delete env; return fail(PMX_ERR_INVALID, "layout_index entry %d outside [0,%d)", v, n_layouts); }
    }
    const int mask = (cfg->obs_agents & 0xF) ? (cfg->obs_agents & 0xF) : 0xF;
    env->cfg.obs_agents = mask;
    env->n_emit = 0;
    for (int i = 0; i < 4; ++i)
        if ((mask >> i) & 1) env->emit[env->n_emit++] = i;
    env->elem_bytes = cfg->obs_dtype == PMX_OBS_F32 ? 4 : (cfg->obs_dtype == PMX_OBS_BF16 ? 2 : 1);
    env->lay_dev = nullptr; env->dump_dev = nullptr; env->state_dev = nullptr; env->snap_dev = nullptr;
    env->layout_idx_dev = nullptr; env->dist_dev = nullptr; env->cell_index_dev = nullptr;
    env->profiling = false; env->ev_rule_used = env->ev_expand_used = 0;

    hipError_t e = hipSetDevice(cfg->device);
    const size_t N = (size_t)cfg->n_envs;
    std::vector<PmxLayoutDev> recs;
    for (auto &l : env->layouts) recs.push_back(l.dev);
    if (e == hipSuccess) e = hipMalloc((void **)&env->lay_dev, sizeof(PmxLayoutDev) * recs.size());
    if (e == hipSuccess) e = hipMalloc((void **)&env->dump_dev, dump.size() + 8);   // read as aligned 32-bit words
    if (e == hipSuccess) e = hipMalloc((void **)&env->state_dev, PMX_STATE_WORDS(H) * N * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&env->snap_dev, 3 * (size_t)PMX_SNAP_WORDS(H) * N * sizeof(uint32_t));
    if (e == hipSuccess && n_layouts > 1) e = hipMalloc((void **)&env->layout_idx_dev, N * sizeof(int32_t));
    if (e == hipSuccess) e = hipMemset(env->state_dev, 0, PMX_STATE_WORDS(H) * N * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemcpy(env->lay_dev, recs.data(), sizeof(PmxLayoutDev) * recs.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(env->dump_dev, dump.data(), dump.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess && n_layouts > 1)
        e = hipMemcpy(env->layout_idx_dev, env->layout_index.data(), N * sizeof(int32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess && with_bots) {
        // maze distances of every layout stay resident for the baselineTeam action codes (distanceCalculator.py:111-150)
        std::vector<int16_t> idx_all;
        std::vector<int8_t> cells_all;
        for (auto &l : env->layouts) idx_all.insert(idx_all.end(), l.cell_index.begin(), l.cell_index.end());
        e = hipMalloc((void **)&env->dist_dev, std::max<size_t>(dist_bytes, 1));
        if (e == hipSuccess) e = hipMalloc((void **)&env->cell_index_dev, idx_all.size() * sizeof(int16_t));
        if (e == hipSuccess) e = hipMemcpy(env->cell_index_dev, idx_all.data(), idx_all.size() * sizeof(int16_t), hipMemcpyHostToDevice);
        int8_t *cells_dev = nullptr;
        if (e == hipSuccess) e = hipMalloc((void **)&cells_dev, 2 * 1024);
        for (size_t li = 0; e == hipSuccess && li < env->layouts.size(); ++li) {
            const pmx_layout_host &lh = env->layouts[li];
            e = hipMemcpy(cells_dev, lh.cells.data(), lh.cells.size(), hipMemcpyHostToDevice);
            if (e == hipSuccess)
                e = pmx_launch_maze(env->lay_dev + li, env->cell_index_dev + li * 1024, lh.dev.n_cells, cells_dev,
                                    env->dist_dev + lh.dev.dist_off, nullptr);
            if (e == hipSuccess) e = hipDeviceSynchronize();
        }
        if (cells_dev) (void)hipFree(cells_dev);
    }
    if (e != hipSuccess) {
        pmx_destroy(env);
        return fail(e == hipErrorOutOfMemory ? PMX_ERR_NOMEM : PMX_ERR_HIP, "pmx_create: %s", hipGetErrorString(e));
    }
    int rc = pmx_reset(env, nullptr, nullptr, nullptr);
    if (rc == PMX_OK && hipDeviceSynchronize() != hipSuccess) rc = fail(PMX_ERR_HIP, "pmx_create: initial reset failed");
    if (rc != PMX_OK) { pmx_destroy(env); return rc; }
    *out = env;
    return PMX_OK;
}

int pmx_destroy(pmx_env *env)
{
    if (!env) return PMX_OK;
    if (env->lay_dev) (void)hipFree(env->lay_dev);
    if (env->dump_dev) (void)hipFree(env->dump_dev);
    if (env->state_dev) (void)hipFree(env->state_dev);
    if (env->snap_dev) (void)hipFree(env->snap_dev);
    if (env->layout_idx_dev) (void)hipFree(env->layout_idx_dev);
    if (env->dist_dev) (void)hipFree(env->dist_dev);
    if (env->cell_index_dev) (void)hipFree(env->cell_index_dev);
    for (hipEvent_t e : env->ev_rule) (void)hipEventDestroy(e);
    for (hipEvent_t e : env->ev_expand) (void)hipEventDestroy(e);
    delete env;
    return PMX_OK;
}

int pmx_obs_shape(const pmx_env *env, int32_t *n_emit, int32_t *height, int32_t *width, int32_t *elem_bytes)
{
    if (!env) return fail(PMX_ERR_INVALID, "null env");
    if (n_emit) *n_emit = env->n_emit;
    if (height) *height = env->lay.H;
    if (width) *width = env->lay.W;
    if (elem_bytes) *elem_bytes = env->elem_bytes;
    return PMX_OK;
}

int pmx_reset(pmx_env *env, const uint8_t *mask_dev, const pmx_step_out *out, void *stream)
{
    if (!env) return fail(PMX_ERR_INVALID, "null env");
    env->snaps_valid = false;      // the sub-step snapshots no longer describe the current state
    PmxTickParams p;
    fill_tick_params(env, p, nullptr, nullptr);
    p.reset_mask = mask_dev;
    p.legal = out ? out->legal_dev : nullptr;
    HIP_TRY(pmx_launch_reset(&p, env->lay.H, as_stream(stream)));
    env->open_agent = 0;
    if (out && out->obs_dev) return launch_expand(env, out->obs_dev, false, -1, as_stream(stream));
    return PMX_OK;
}

int pmx_step(pmx_env *env, const int8_t *actions_dev, const pmx_step_out *out, void *stream)
{
    if (!env || !actions_dev) return fail(PMX_ERR_INVALID, "pmx_step: null argument");
    if (env->open_agent != 0) return fail(PMX_ERR_INVALID, "pmx_step: a tick opened with pmx_step_agent is unfinished");
    PmxTickParams p;
    fill_tick_params(env, p, actions_dev, out);
    hipEvent_t *ev = prof_pair(env, false);
    HIP_TRY(pmx_launch_rule(&p, env->lay.H, as_stream(stream), ev ? ev[0] : nullptr, ev ? ev[1] : nullptr));
    env->snaps_valid = true;
    if (out && out->obs_dev) return launch_expand(env, out->obs_dev, true, -1, as_stream(stream));
    return PMX_OK;
}

int pmx_emit_team_obs(pmx_env *env, int team_red, void *team_obs_dev, void *merged_dev, void *stream)
{
    if (!env || !team_obs_dev) return fail(PMX_ERR_INVALID, "pmx_emit_team_obs: null argument");
    if (env->open_agent != 0) return fail(PMX_ERR_INVALID, "pmx_emit_team_obs: a tick opened with pmx_step_agent is unfinished");
    PmxEmitParams x;
    std::memset(&x, 0, sizeof(x));
    const size_t snap_sz = (size_t)PMX_SNAP_WORDS(env->lay.H) * env->cfg.n_envs;
    for (int a = 0; a < 4; ++a) x.snap[a] = (env->snaps_valid && a < 3) ? env->snap_dev + a * snap_sz : env->state_dev;
    x.lay = env->lay_dev;
    x.layout_idx = env->layout_idx_dev;
    x.team_obs = team_obs_dev;
    x.merged = merged_dev;
    x.N = env->cfg.n_envs;
    x.red = team_red ? 1 : 0;
    x.lay_H = env->lay.H, x.lay_W = env->lay.W;
    // profiling: an emit launch is timed like an expansion launch (it takes the expansion's place in a training loop, whose
    // pmx_step calls pass no observation pointer and therefore launch no expansion of their own)
    hipEvent_t *ev = prof_pair(env, true);
    HIP_TRY(pmx_launch_emit_team(&x, env->cfg.obs_dtype, as_stream(stream), ev ? ev[0] : nullptr, ev ? ev[1] : nullptr));
    return PMX_OK;
}

int pmx_step_agent(pmx_env *env, int agent, const int8_t *actions_dev, const pmx_step_out *out, void *stream)
{
    if (!env || !actions_dev) return fail(PMX_ERR_INVALID, "pmx_step_agent: null argument");
    env->snaps_valid = false;      // the sub-step snapshots no longer describe the current state
    if (agent != env->open_agent)
        return fail(PMX_ERR_INVALID, "pmx_step_agent: expected agent %d, got %d (sub-steps run 0,1,2,3)", env->open_agent, agent);
    PmxTickParams p;
    fill_tick_params(env, p, actions_dev, out);   // the kernel writes the tick-level outputs only when agent == 3
    HIP_TRY(pmx_launch_rule_agent(&p, env->lay.H, agent, as_stream(stream)));
    env->open_agent = (agent + 1) & 3;
    if (out && out->obs_dev) return launch_expand(env, out->obs_dev, false, agent, as_stream(stream));
    return PMX_OK;
}

int pmx_successor(pmx_env *env, int agent, const int8_t *actions_dev, int32_t *score_change_dev, void *stream)
{
    if (!env || !actions_dev) return fail(PMX_ERR_INVALID, "pmx_successor: null argument");
    env->snaps_valid = false;      // the sub-step snapshots no longer describe the current state
    if (agent < 0 || agent > 3) return fail(PMX_ERR_INVALID, "pmx_successor: agent %d out of range", agent);
    PmxTickParams p;
    fill_tick_params(env, p, actions_dev, nullptr);
    p.score_change = score_change_dev;
    HIP_TRY(pmx_launch_successor(&p, env->lay.H, agent, as_stream(stream)));
    return PMX_OK;
}

int pmx_observe(pmx_env *env, void *obs_dev, uint8_t *legal_dev, void *stream)
{
    if (!env) return fail(PMX_ERR_INVALID, "null env");
    if (legal_dev) {
        PmxTickParams p;
        fill_tick_params(env, p, nullptr, nullptr);
        p.legal = legal_dev;
        p.no_reset = 1;
        HIP_TRY(pmx_launch_reset(&p, env->lay.H, as_stream(stream)));
    }
    if (obs_dev) return launch_expand(env, obs_dev, false, -1, as_stream(stream));
    return PMX_OK;
}

// ---- per-kernel timing for bench.py (not part of the reference surface) -----------------------------------------
// Launch tuning of the expansion kernel for A/B measurements: key "expand_alt" | "expand_nt" | "expand_lds_pad" | "expand_lut",
// value -1 = built-in choice.
int pmx_set_tuning(pmx_env *env, const char *key, int32_t value)
{
    if (!env || !key) return fail(PMX_ERR_INVALID, "pmx_set_tuning: null argument");
    if (!strcmp(key, "expand_alt")) env->tune.alt = value;
    else if (!strcmp(key, "expand_nt")) env->tune.nt = value;
    else if (!strcmp(key, "expand_lds_pad")) env->tune.lds_pad = value;
    else if (!strcmp(key, "expand_lut")) env->tune.lut = value;
    else if (!strcmp(key, "expand_wave_per_env")) env->tune.per_env = value;
    else return fail(PMX_ERR_INVALID, "pmx_set_tuning: unknown key %s", key);
    return PMX_OK;
}

// Between begin and end every pmx_step / pmx_observe / pmx_reset records a HIP event pair around its rule-kernel and
// expansion-kernel launches, on the stream the kernels run on.  pmx_profile_end synchronises those events and returns
// the summed kernel durations in milliseconds and the launch counts.
int pmx_profile_begin(pmx_env *env, int32_t max_launches)
{
    if (!env || max_launches < 1) return fail(PMX_ERR_INVALID, "pmx_profile_begin: bad argument");
    auto grow = [&](std::vector<hipEvent_t> &v) -> hipError_t {
        while (v.size() < (size_t)max_launches * 2) {
            hipEvent_t e;
            hipError_t rc = hipEventCreate(&e);
            if (rc != hipSuccess) return rc;
            v.push_back(e);
        }
        return hipSuccess;
    };
    HIP_TRY(grow(env->ev_rule));
    HIP_TRY(grow(env->ev_expand));
    env->ev_rule_used = env->ev_expand_used = 0;
    env->profiling = true;
    return PMX_OK;
}

int pmx_profile_end(pmx_env *env, double *rule_ms, int32_t *rule_launches, double *expand_ms, int32_t *expand_launches)
{
    if (!env) return fail(PMX_ERR_INVALID, "null env");
    env->profiling = false;
    auto sum = [&](std::vector<hipEvent_t> &v, size_t used, double *ms, int32_t *n) -> hipError_t {
        double acc = 0.0;
        for (size_t k = 0; k + 1 < used; k += 2) {
            hipError_t rc = hipEventSynchronize(v[k + 1]);
            if (rc != hipSuccess) return rc;
            float t = 0.f;
            rc = hipEventElapsedTime(&t, v[k], v[k + 1]);
            if (rc != hipSuccess) return rc;
            acc += t;
        }
        if (ms) *ms = acc;
        if (n) *n = (int32_t)(used / 2);
        return hipSuccess;
    };
    HIP_TRY(sum(env->ev_rule, env->ev_rule_used, rule_ms, rule_launches));
    HIP_TRY(sum(env->ev_expand, env->ev_expand_used, expand_ms, expand_launches));
    return PMX_OK;
}

// ---- host <-> device state exchange --------------------------------------------------------------------------

int pmx_get_state(pmx_env *env, int32_t first, int32_t count, pmx_state *states, void *stream)
{
    if (!env || !states) return fail(PMX_ERR_INVALID, "pmx_get_state: null argument");
    const int N = env->cfg.n_envs, H = env->lay.H;
    if (first < 0 || count < 0 || first + count > N) return fail(PMX_ERR_INVALID, "pmx_get_state: range outside [0,%d)", N);
    if (count == 0) return PMX_OK;
    const int words = PMX_STATE_WORDS(H);
    std::vector<uint32_t> buf((size_t)words * count);
    HIP_TRY(hipMemcpy2DAsync(buf.data(), (size_t)count * 4, env->state_dev + first, (size_t)N * 4, (size_t)count * 4, words,
                             hipMemcpyDeviceToHost, as_stream(stream)));
    HIP_TRY(hipStreamSynchronize(as_stream(stream)));
    for (int k = 0; k < count; ++k) {
        pmx_state &s = states[k];
        std::memset(&s, 0, sizeof(s));
        auto word = [&](int w) { return buf[(size_t)w * count + k]; };
        for (int y = 0; y < H; ++y) s.food[y] = word(y);
        for (int i = 0; i < 4; ++i) {
            const uint32_t a = word(PMX_W_AGENT_A(H, i)), b = word(PMX_W_AGENT_B(H, i));
            s.pos[i][0] = (int8_t)(a & 0xFF); s.pos[i][1] = (int8_t)((a >> 8) & 0xFF);
            s.dir[i] = (int8_t)((a >> 16) & 0xFF); s.pac[i] = (uint8_t)((a >> 24) & 1);
            s.scared[i] = (uint8_t)(b & 0xFF); s.carry[i] = (uint16_t)((b >> 8) & 0xFFF); s.ret[i] = (uint16_t)((b >> 20) & 0xFFF);
        }
        for (int j = 0; j < 4; ++j) {
            const uint32_t c = (word(PMX_W_CAPS(H, j >> 1)) >> (16 * (j & 1))) & 0xFFFFu;
            if (c != 0xFFFFu) s.caps[c >> 8] |= 1u << (c & 0xFF);
        }
        s.score = (int32_t)word(PMX_W_SCORE(H));
        s.steps = (int32_t)word(PMX_W_STEPS(H));
        s.ticks = word(PMX_W_TICKS(H));
    }
    return PMX_OK;
}

// with redraw_layouts the device owns the env -> layout map: refresh the host copy before it is used
static int refresh_layout_index(pmx_env *env, hipStream_t st)
{
    if (!env->cfg.redraw_layouts || !env->layout_idx_dev) return PMX_OK;
    HIP_TRY(hipMemcpyAsync(env->layout_index.data(), env->layout_idx_dev, env->layout_index.size() * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return PMX_OK;
}

int pmx_get_layout_index(pmx_env *env, int32_t *index_out, void *stream)
{
    if (!env || !index_out) return fail(PMX_ERR_INVALID, "pmx_get_layout_index: null argument");
    if (env->layout_index.empty()) {
        std::fill(index_out, index_out + env->cfg.n_envs, 0);
        return PMX_OK;
    }
    if (int rc = refresh_layout_index(env, as_stream(stream))) return rc;
    std::copy(env->layout_index.begin(), env->layout_index.end(), index_out);
    return PMX_OK;
}

int pmx_set_state(pmx_env *env, int32_t first, int32_t count, const pmx_state *states, void *stream)
{
    if (!env || !states) return fail(PMX_ERR_INVALID, "pmx_set_state: null argument");
    env->snaps_valid = false;      // the sub-step snapshots no longer describe the current state
    const int N = env->cfg.n_envs, H = env->lay.H, W = env->lay.W;
    if (first < 0 || count < 0 || first + count > N) return fail(PMX_ERR_INVALID, "pmx_set_state: range outside [0,%d)", N);
    if (count == 0) return PMX_OK;
    if (int rc = refresh_layout_index(env, as_stream(stream))) return rc;
    const int words = PMX_STATE_WORDS(H);
    std::vector<uint32_t> buf((size_t)words * count, 0);
    for (int k = 0; k < count; ++k) {
        const pmx_state &s = states[k];
        const PmxLayoutDev &lay = env->layouts[env->layout_index.empty() ? 0 : env->layout_index[first + k]].dev;
        auto word = [&](int w) -> uint32_t & { return buf[(size_t)w * count + k]; };
        uint16_t slots[4] = { 0xFFFF, 0xFFFF, 0xFFFF, 0xFFFF };
        int nc = 0;
        for (int y = 0; y < H; ++y) {
            if ((s.food[y] | s.caps[y]) & lay.walls[y]) return fail(PMX_ERR_INVALID, "state %d: food/capsule inside a wall (row %d)", k, y);
            word(y) = s.food[y];
            for (int x = 0; x < W; ++x)
                if ((s.caps[y] >> x) & 1u) {
                    if (nc == PMX_MAX_CAPSULES) return fail(PMX_ERR_UNSUPPORTED, "state %d: more than %d capsules", k, PMX_MAX_CAPSULES);
                    slots[nc++] = (uint16_t)(x | (y << 8));
                }
        }
        for (int i = 0; i < 4; ++i) {
            const int x = s.pos[i][0], y = s.pos[i][1];
            if (x <= 0 || y <= 0 || x >= W - 1 || y >= H - 1 || ((lay.walls[y] >> x) & 1u))
                return fail(PMX_ERR_INVALID, "state %d: agent %d at (%d,%d) is not on an open interior cell", k, i, x, y);
            if (s.dir[i] < 0 || s.dir[i] > 4 || s.carry[i] > 0xFFF || s.ret[i] > 0xFFF)
                return fail(PMX_ERR_INVALID, "state %d: agent %d field out of range", k, i);
            word(PMX_W_AGENT_A(H, i)) = (uint32_t)x | ((uint32_t)y << 8) | ((uint32_t)s.dir[i] << 16) | ((uint32_t)(s.pac[i] != 0) << 24);
            word(PMX_W_AGENT_B(H, i)) = (uint32_t)s.scared[i] | ((uint32_t)s.carry[i] << 8) | ((uint32_t)s.ret[i] << 20);
        }
        word(PMX_W_CAPS(H, 0)) = slots[0] | ((uint32_t)slots[1] << 16);
        word(PMX_W_CAPS(H, 1)) = slots[2] | ((uint32_t)slots[3] << 16);
        word(PMX_W_SCORE(H)) = (uint32_t)s.score;
        word(PMX_W_STEPS(H)) = (uint32_t)s.steps;
        word(PMX_W_TICKS(H)) = s.ticks;
    }
    HIP_TRY(hipMemcpy2DAsync(env->state_dev + first, (size_t)N * 4, buf.data(), (size_t)count * 4, (size_t)count * 4, words,
                             hipMemcpyHostToDevice, as_stream(stream)));
    HIP_TRY(hipStreamSynchronize(as_stream(stream)));
    env->open_agent = 0;
    return PMX_OK;
}

int pmx_maze_distances_layout(pmx_env *env, int32_t layout, int8_t *cells_dev, uint8_t *dist_dev, int32_t *n_cells, void *stream)
{
    if (!env) return fail(PMX_ERR_INVALID, "null env");
    if (layout < 0 || layout >= (int32_t)env->layouts.size()) return fail(PMX_ERR_INVALID, "layout %d out of range", layout);
    const pmx_layout_host &lh = env->layouts[layout];
    const int n = (int)(lh.cells.size() / 2);
    if (n_cells) *n_cells = n;
    if (cells_dev) HIP_TRY(hipMemcpyAsync(cells_dev, lh.cells.data(), lh.cells.size(), hipMemcpyHostToDevice, as_stream(stream)));
    if (!dist_dev) {
        if (cells_dev) HIP_TRY(hipStreamSynchronize(as_stream(stream)));
        return PMX_OK;
    }
    if (!cells_dev) return fail(PMX_ERR_INVALID, "pmx_maze_distances: cells_dev is required with dist_dev");
    int16_t *idx_dev = nullptr;
    HIP_TRY(hipMallocAsync((void **)&idx_dev, lh.cell_index.size() * sizeof(int16_t), as_stream(stream)));
    HIP_TRY(hipMemcpyAsync(idx_dev, lh.cell_index.data(), lh.cell_index.size() * sizeof(int16_t), hipMemcpyHostToDevice,
                           as_stream(stream)));
    hipError_t e = pmx_launch_maze(env->lay_dev + layout, idx_dev, n, cells_dev, dist_dev, as_stream(stream));
    // the host vectors outlive the async copies only if we wait for them here
    HIP_TRY(hipStreamSynchronize(as_stream(stream)));
    HIP_TRY(hipFreeAsync(idx_dev, as_stream(stream)));
    if (e != hipSuccess) return fail(PMX_ERR_HIP, "pmx_maze_distances: %s", hipGetErrorString(e));
    return PMX_OK;
}

int pmx_maze_distances(pmx_env *env, int8_t *cells_dev, uint8_t *dist_dev, int32_t *n_cells, void *stream)
{
    return pmx_maze_distances_layout(env, 0, cells_dev, dist_dev, n_cells, stream);
}

}  // extern "C"
