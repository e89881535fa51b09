/*
 * pmx.h -- C ABI of the MI355X-native Pac-Man Capture-the-Flag environment step.
 *
 * The reference (ceselder/pacman-marl-2025) is plain Python and has no FFI of its own; the boundary it
 * offers for this path is the call surface of gymPacMan.gymPacMan_parallel_env (gymPacMan.py:15-270) and
 * the CaptureAgent action API (captureAgents.py:91-162).  This header is the C ABI a maintainer would bind
 * UNDER those Python classes (ctypes stub in INTEGRATION.md); each entry point cites what it replaces.
 *
 * Conventions
 *   - every function returns 0 on success or a negative PMX_ERR_* code; pmx_last_error() returns a
 *     thread-local description of the last failure.  No exceptions cross the boundary.
 *   - the caller owns every buffer; the library owns only the opaque pmx_env handle (and the device
 *     memory behind it).  Pointers named *_dev are DEVICE pointers (HIP), everything else is host memory.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  All *_dev work is
 *     stream-ordered and asynchronous; the library never synchronises unless the comment says so.
 *   - a handle is bound to one device and is not thread-safe.
 *   - coordinates: (x, y), origin bottom-left; bit x of row word y (game.py:162-167, layout.py:108-112).
 *   - actions / directions: 0 North, 1 East, 2 South, 3 West, 4 Stop (gymPacMan.py:66-89).
 *   - agents 0,2 are red and start on the left half, agents 1,3 are blue (capture.py:316-319,
 *     gymPacMan.py:150,185,210); pmx_create rejects layouts where that does not hold.
 */
#ifndef PMX_H
#define PMX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PMX_VERSION 1
#define PMX_MAX_DIM 32      /* width and height <= 32: one uint32 per board row */
#define PMX_MAX_CAPSULES 4

enum {
    PMX_OK = 0,
    PMX_ERR_INVALID = -1,       /* bad argument / malformed layout */
    PMX_ERR_UNSUPPORTED = -2,   /* valid request this build does not implement */
    PMX_ERR_HIP = -3,           /* a HIP runtime call failed (message has the HIP error string) */
    PMX_ERR_NOMEM = -4
};

enum { PMX_OBS_F32 = 0, PMX_OBS_BF16 = 1, PMX_OBS_U8 = 2 };

/* Action code for an in-kernel randomTeam opponent (agents/randomTeam.py:90-100: random.choice(legal actions)):
 * the env itself picks uniformly among the agent's legal actions ON THE MID-TICK STATE, in the reference's list order
 * N,S,E,W,Stop, with a counter-based generator keyed by (config seed, env index, tick counter, agent).  The draw is
 * distribution-equivalent to the reference's (which uses Python's global Mersenne Twister) and is reproduced exactly
 * by the test oracle. */
#define PMX_ACTION_RANDOM_LEGAL (-2)
/* In-kernel baselineTeam opponents (agents/baselineTeam.py:65-187): the env scores the successor of every legal action of
 * the agent on the mid-tick state with the reference's reflex features and weights (offensive: -100 * food left - maze
 * distance to the nearest pellet; defensive: invaders, on-defence, invader distance, stop, reverse) and plays one of the
 * best actions (tie-break by the counter-based generator instead of random.choice); with no food left the agent walks home.
 * Only handles created with pmx_config.enable_bots understand these codes (the layouts' maze-distance matrices then stay
 * resident on the device, at most 2 GiB); elsewhere they act as Stop. */
#define PMX_ACTION_BASELINE_OFFENSE (-3)
#define PMX_ACTION_BASELINE_DEFENSE (-4)

typedef struct pmx_env pmx_env;

/* What gymPacMan_parallel_env.__init__ takes (gymPacMan.py:15) plus the batch size.  The .lay text is parsed
 * by the host language (layout.py:95-130); rows arrive bottom-up as bit masks. */
typedef struct {
    int32_t width, height;
    const uint32_t *wall_rows;   /* [height] '%' cells */
    const uint32_t *food_rows;   /* [height] '.' cells */
    const uint32_t *cap_rows;    /* [height] 'o' cells (at most PMX_MAX_CAPSULES bits) */
    const int8_t *starts;        /* [4][2] (x, y) of layout digits 1..4 = agents 0..3 (layout.py:113-114,128-129) */
    int32_t n_envs;              /* independent games advanced in lock-step */
    int32_t length;              /* `length` (gymPacMan.py:55): an episode lasts length+1 ticks (gymPacMan.py:268) */
    int32_t legal_reward;        /* reward_forLegalAction (gymPacMan.py:254-257) */
    int32_t defence_reward;      /* defenceReward (gymPacMan.py:238-242) */
    int32_t auto_reset;          /* 1: an env that terminates is reset inside the step, the way the reference's
                                    caller does right after a done (pacman_mappo_resnet.py:528-538); the
                                    observations/legal masks returned for it are those of the fresh game */
    int32_t obs_dtype;           /* PMX_OBS_*: element type of the observation planes (reference: float32) */
    int32_t obs_agents;          /* bit i set = emit agent i's observation; 0 means all four (0xF) */
    int32_t device;              /* HIP device ordinal */
    uint32_t seed;               /* key of the PMX_ACTION_RANDOM_LEGAL generator */
    int32_t n_layouts;           /* 0 or 1: every env plays the one layout above.  L > 1: wall_rows / food_rows / cap_rows
                                    hold [L][height] rows and starts [L][4][2]; all layouts share width x height; an env
                                    keeps the layout it is given unless redraw_layouts is set (below) */
    const int32_t *layout_index; /* [n_envs] host array: layout of each env (required when n_layouts > 1) */
    int32_t enable_bots;         /* 1: keep every layout's maze-distance matrix on the device and use the tick kernel variant
                                    that understands PMX_ACTION_BASELINE_* (it is a few microseconds per tick slower) */
    int32_t redraw_layouts;      /* 1 (needs n_layouts > 1): random_layout=True of the reference (gymPacMan.py:98-100 draws a new
                                    maze at every reset()): whenever an env is reset -- by pmx_reset or by the auto-reset inside
                                    pmx_step -- it moves to layout floor(u * n_layouts) of the pool, u from the counter-based
                                    generator keyed by (seed, env index, the env's tick counter); the layouts are the pool the
                                    host generated (maze_generator.py), layout_index gives the initial assignment.  The draw is
                                    distribution-equivalent to the reference's (which reseeds Python's global generator per maze)
                                    and is reproduced exactly by the test oracle. */
} pmx_config;

/* Outputs of one tick = what gymPacMan.step returns (gymPacMan.py:191-193), batched.  Any pointer may be
 * NULL to skip that output. */
typedef struct {
    void *obs_dev;               /* [n_envs][n_emit][8][H][W] obs_dtype; n_emit = popcount(obs_agents), agents in
                                    increasing index order.  Agent i's planes are encoded right after agent i's own
                                    sub-step (gymPacMan.py:166-167) */
    double *reward_dev;          /* [n_envs][2] team reward (red, blue) as float64, summed in the reference's order */
    uint8_t *done_dev;           /* [n_envs] terminations (gymPacMan.py:261-270) */
    uint8_t *legal_dev;          /* [n_envs][4] bit a = action a legal in the state the caller acts on next */
    int32_t *score_change_dev;   /* [n_envs] info['score_change'] */
    int32_t *score_dev;          /* [n_envs] game.state.data.score after the tick, BEFORE any auto-reset
                                    (pacman_mappo_resnet.py:529 reads it at a done) */
    uint32_t *agent_dev;         /* [n_envs][4] x | y << 8 | numCarrying << 16 of each agent right after its OWN sub-step,
                                    before any auto-reset: the content of observation plane 1 in compact form, which
                                    is all that get_agent_state / compute_heuristic_shaping read
                                    (pacman_mappo_resnet.py:241-264) */
} pmx_step_out;

/* Dynamic part of one game = the fields of capture.GameState / game.AgentState that the path reads
 * (game.py:120-131,374-396).  Used by pmx_get_state / pmx_set_state, fixtures and the GameState facade. */
typedef struct {
    int8_t pos[4][2];
    int8_t dir[4];               /* configuration.direction */
    uint8_t pac[4];              /* isPacman */
    uint8_t scared[4];           /* scaredTimer */
    uint16_t carry[4];           /* numCarrying */
    uint16_t ret[4];             /* numReturned */
    uint32_t food[PMX_MAX_DIM];
    uint32_t caps[PMX_MAX_DIM];
    int32_t score;
    int32_t steps;               /* gymPacMan_parallel_env.steps */
    uint32_t ticks;              /* ticks since pmx_create (never reset): counter of the random-legal generator */
} pmx_state;

int pmx_version(void);
const char *pmx_last_error(void);

/* gymPacMan_parallel_env.__init__ (gymPacMan.py:15-89) + CaptureRules.newGame (capture.py:369-382): validates the
 * layout, allocates device state for n_envs games and puts every game in its initial state. */
int pmx_create(const pmx_config *cfg, pmx_env **out);
int pmx_destroy(pmx_env *env);

/* shape helpers for the binding */
int pmx_obs_shape(const pmx_env *env, int32_t *n_emit, int32_t *height, int32_t *width, int32_t *elem_bytes);

/* gymPacMan_parallel_env.reset (gymPacMan.py:92-141).  mask_dev: [n_envs] non-zero = reset that env; NULL = all.
 * out (may be NULL): obs_dev / legal_dev are filled for ALL envs from their current state (the four agents see
 * the same state, gymPacMan.py:135-137); the other members are ignored. */
int pmx_reset(pmx_env *env, const uint8_t *mask_dev, const pmx_step_out *out, void *stream);

/* gymPacMan_parallel_env.step with self_play=True (gymPacMan.py:143-193): actions_dev [n_envs][4] int8, one
 * requested action per agent; an illegal or out-of-range action becomes Stop (capture.py:473-474). */
int pmx_step(pmx_env *env, const int8_t *actions_dev, const pmx_step_out *out, void *stream);

/* One agent's sub-step of the same tick (the body of the loop gymPacMan.py:149-169), so that host-side
 * CaptureAgent bots can choose on the mid-tick state (gymPacMan.py:157).  Call with agent = 0,1,2,3 in this
 * order; actions_dev [n_envs] int8.  out->obs_dev, if set, receives [n_envs][8][H][W] for this agent.  The
 * call with agent == 3 closes the tick: reward/done/legal/score_change/score are written then (and only then),
 * and auto-reset, if configured, is applied. */
int pmx_step_agent(pmx_env *env, int agent, const int8_t *actions_dev, const pmx_step_out *out, void *stream);

/* GameState.generateSuccessor (capture.py:107-123) applied in place to every env: agent `agent` takes actions_dev[env]
 * (int8 [n_envs]).  No reward, no tick bookkeeping, no termination test: this is the query CaptureAgent bots make on
 * hypothetical states (agents/baselineTeam.py:94-104), normally on a small scratch handle loaded with pmx_set_state.
 * score_change_dev (may be NULL) receives data.scoreChange of each successor. */
int pmx_successor(pmx_env *env, int agent, const int8_t *actions_dev, int32_t *score_change_dev, void *stream);

/* Observation planes / legal masks of the CURRENT state for all emitted agents (gymPacMan.get_Observation,
 * gymPacMan.py:195-229; GameState.getLegalActions, capture.py:101-105). */
int pmx_observe(pmx_env *env, void *obs_dev, uint8_t *legal_dev, void *stream);

/* What the MAPPO rollout does with every tick's observations (pacman_mappo_resnet.py:462-469), straight from the env: the two
 * observations of one team -- agents (0, 2) when team_red, else (1, 3); agent i's planes from the state right after its own
 * sub-step, exactly as pmx_step emits them -- written as team_obs_dev [n_envs][2][8][H][W] of the handle's obs_dtype and
 * CANONICALISED for a red team (canonicalize_obs :215-229: x flipped, planes 2 <-> 3 and 6 <-> 7 swapped), plus, if merged_dev
 * is not NULL, merge_obs_for_critic of the two (:267-274) as [n_envs][8][H][W].  Valid after pmx_step (tick observations) and
 * after pmx_reset / pmx_set_state (all agents see the current state). */
int pmx_emit_team_obs(pmx_env *env, int team_red, void *team_obs_dev, void *merged_dev, void *stream);

/* The layout each env is currently on (host output [n_envs]; all zeros for a single-layout handle).  Synchronises the stream. */
int pmx_get_layout_index(pmx_env *env, int32_t *index_out, void *stream);

/* Host copies of `count` games starting at `first`.  These two calls synchronise the stream. */
int pmx_get_state(pmx_env *env, int32_t first, int32_t count, pmx_state *states, void *stream);
int pmx_set_state(pmx_env *env, int32_t first, int32_t count, const pmx_state *states, void *stream);

/* distanceCalculator.computeDistances (distanceCalculator.py:111-150) for the env's layout.  cells_dev
 * [n_cells][2] int8 receives the open cells in Grid.asList(False) order (game.py:225-230), dist_dev
 * [n_cells][n_cells] uint8 the shortest 4-neighbour path lengths (255 = unreachable).  *n_cells is a host
 * output available on return (the count is computed on the host); dist_dev may be NULL to query it. */
int pmx_maze_distances(pmx_env *env, int8_t *cells_dev, uint8_t *dist_dev, int32_t *n_cells, void *stream);
/* the same for layout `layout` of a multi-layout handle */
int pmx_maze_distances_layout(pmx_env *env, int32_t layout, int8_t *cells_dev, uint8_t *dist_dev, int32_t *n_cells, void *stream);

/* Measurement hooks (no reference counterpart): between begin and end, every tick attaches a start and a stop HIP event
 * to its rule-kernel and to its expansion-kernel dispatch on the caller's stream (hipExtLaunchKernelGGL); end
 * synchronises them and returns the summed kernel durations (milliseconds) and launch counts.  bench.py's roofline
 * figures come from here.  A pmx_emit_team_obs launch is recorded with the expansion launches (a training loop calls it in
 * place of the four-agent expansion). */
int pmx_profile_begin(pmx_env *env, int32_t max_launches);
/* Launch tuning of the observation-expansion kernel for A/B measurements (no reference counterpart).  key: "expand_alt"
 * (1 = walk the planes in alternating directions from tick to tick, the default; 0 = always the same direction, so that every
 * byte reaches HBM even when the caller re-steps one buffer that partly fits the Infinity Cache), "expand_nt" (non-temporal
 * stores), "expand_lds_pad" (occupancy cap, bytes of dynamic LDS), "expand_lut", "expand_wave_per_env" (one wave per env instead of
 * one per (env, agent)).  value -1 restores the built-in choice.  The same settings are read from the environment variables
 * PMX_EXPAND_ALT / _NT / _LDS_PAD / _LUT / _PER_ENV once, at pmx_create. */
int pmx_set_tuning(pmx_env *env, const char *key, int32_t value);
int pmx_profile_end(pmx_env *env, double *rule_ms, int32_t *rule_launches, double *expand_ms, int32_t *expand_launches);

/* pacman_mappo_resnet.compute_gae (pacman_mappo_resnet.py:277-290) for n independent series laid out
 * [T][n] (time-major): rewards/values/dones float32, last_value [n] float32; adv/ret [T][n] float32. */
int pmx_gae(const float *rewards_dev, const float *values_dev, const float *dones_dev, const float *last_value_dev,
            int32_t T, int32_t n, double gamma, double lam, float *adv_dev, float *ret_dev, void *stream);

/* Column sums of a [rows][C] bfloat16 matrix as PMX_COLSUM_BLOCKS partial rows of float32 (the caller adds them): the
 * bias gradients `grad_output.sum((0, 1))` of the token linears in the critic's encoder layers (nn.TransformerEncoderLayer
 * backward, pacman_mappo_resnet.py:138-141).  C a multiple of 8, at most 256. */
#define PMX_COLSUM_BLOCKS 512
int pmx_colsum_bf16(const void *x_dev, int64_t rows, int32_t C, float *partial_dev, void *stream);

/* Training-side observation post-processing on [n][8][H][W] blocks of obs_dtype elements, for callers that take the four
 * observations of pmx_step and post-process them the way the reference script does (pacman_mappo_resnet.py:215-229
 * canonicalize_obs for a red learner, :267-274 merge_obs_for_critic): pmx.trainer.canonicalize_obs / merge_obs on device
 * tensors.  (The training loop itself gets both from pmx_emit_team_obs, already written into its rollout buffers.) */
int pmx_canonicalize_obs(const void *in_dev, void *out_dev, int32_t n, int32_t H, int32_t W, int32_t obs_dtype,
                         void *stream);
int pmx_merge_obs(const void *a_dev, const void *b_dev, void *out_dev, int32_t n, int32_t H, int32_t W,
                  int32_t obs_dtype, void *stream);

/* Fused residual add + LayerNorm over a feature dimension of 32 (the critic's post-LN encoder layers,
 * pacman_mappo_resnet.py:138-141: norm(x + sublayer(x))): y = LayerNorm(x + a) * w + b on [rows][32] tensors of float32
 * (dtype 0) or bfloat16 (dtype 1); mean / rstd [rows] float32 are saved for the backward pass.  The backward pass
 * returns dz = d loss / d (x + a) (the gradient of both inputs) and per-wavefront partial sums of the weight / bias
 * gradients: partial_dev [PMX_LN32_PARTIAL_ROWS][64] float32 (dw in columns 0..31, db in 32..63), fully overwritten;
 * the caller sums the rows. */
#define PMX_LN32_PARTIAL_ROWS 2048
int pmx_ln32_forward(const void *x_dev, const void *a_dev, const float *w_dev, const float *b_dev, void *y_dev, float *mean_dev,
                     float *rstd_dev, int64_t rows, float eps, int32_t dtype, void *stream);
int pmx_ln32_backward(const void *x_dev, const void *a_dev, const void *dy_dev, const float *w_dev, const float *mean_dev,
                      const float *rstd_dev, void *dz_dev, float *partial_dev, int64_t rows, int32_t dtype, void *stream);

/* Fused GroupNorm (8 channels per group) + optional residual add + exact GELU on [B][groups*8][H*W] tensors (NCHW),
 * the elementwise tail of the actor's residual blocks (pacman_mappo_resnet.py:58-67): y = GELU(GN(h) * w + b (+ res)).
 * dtype 0 float32, 1 bfloat16; res_dev / dres_dev may both be NULL.  Backward writes dh (and dres) and
 * partial_dev [B*groups][8][2] float32 = per (sample, group, channel) sums of dz*xhat and dz, which the caller adds up
 * over the batch to get the weight and bias gradients. */
int pmx_gn8_gelu_forward(const void *h_dev, const void *res_dev, const float *w_dev, const float *b_dev, void *y_dev, float *mean_dev,
                         float *rstd_dev, int64_t B, int32_t groups, int32_t HW, float eps, int32_t dtype, void *stream);
int pmx_gn8_gelu_backward(const void *h_dev, const void *res_dev, const void *dy_dev, const float *w_dev, const float *b_dev,
                          const float *mean_dev, const float *rstd_dev, void *dh_dev, void *dres_dev, float *partial_dev, int64_t B,
                          int32_t groups, int32_t HW, int32_t dtype, void *stream);

/* The same for channels-last bfloat16 tensors ([B][H*W][groups*8] in memory: what MIOpen's NHWC convolutions read and write). */
int pmx_gn8cl_gelu_forward(const void *h_dev, const void *res_dev, const float *w_dev, const float *b_dev, void *y_dev, float *mean_dev,
                           float *rstd_dev, int64_t B, int32_t groups, int32_t HW, float eps, void *stream);
int pmx_gn8cl_gelu_backward(const void *h_dev, const void *res_dev, const void *dy_dev, const float *w_dev, const float *b_dev,
                            const float *mean_dev, const float *rstd_dev, void *dh_dev, void *dres_dev, float *partial_dev, int64_t B,
                            int32_t groups, int32_t HW, void *stream);

/* Self-attention forward of the critic's encoder layers (nn.MultiheadAttention, embed 32, 4 heads of 8,
 * pacman_mappo_resnet.py:138-141) on the matrix cores: qkv_dev [S][B][96] bfloat16 = packed in-projection output,
 * out_dev [S][B][32] bfloat16 = concatenated heads before the out-projection, lse_dev [B][4][S] float32 (may be NULL).
 * S <= 1024. */
int pmx_attn8_forward(const void *qkv_dev, void *out_dev, float *lse_dev, int32_t S, int32_t B, void *stream);
/* Its backward pass: dqkv_dev [S][B][96] bfloat16 from the saved qkv, out, lse and the incoming dout [S][B][32].  S <= 640. */
int pmx_attn8_backward(const void *qkv_dev, const void *out_dev, const void *dout_dev, const float *lse_dev, void *dqkv_dev,
                       int32_t S, int32_t B, void *stream);
/* The same two kernels on batch-major tensors when batch_major != 0: qkv [B][S][96], out / dout [B][S][32], dqkv [B][S][96]
 * (lse stays [B][4][S]).  That is the memory order of a channels-last convolution output [B][H][W][32] read as tokens, so the
 * critic (pacman_mappo_resnet.py:160-170: projector -> flatten(2).permute(2, 0, 1) -> encoder) needs no transposing copy
 * between its convolution and its encoder layers, and a sample's rows are contiguous for the kernels. */
int pmx_attn8_forward_layout(const void *qkv_dev, void *out_dev, float *lse_dev, int32_t S, int32_t B, int32_t batch_major, void *stream);
int pmx_attn8_backward_layout(const void *qkv_dev, const void *out_dev, const void *dout_dev, const float *lse_dev, void *dqkv_dev,
                              int32_t S, int32_t B, int32_t batch_major, void *stream);

/* The PPO minibatch objective (pacman_mappo_resnet.py:571-585) and its gradient with respect to the network outputs in one
 * launch: logits_dev [B][5] float32 (logits_bf16 = 0) or bfloat16 (1); values_dev [BV] float32 with BV == B, or BV == B / 2 for
 * paired minibatches (rows 2k, 2k + 1 are the two learners of one env-tick and share value k); act_dev [B] int64; old_logp,
 * adv, ret [B] float32.  clip_eps / ent_coef come from the device pointers when those are not NULL (graph replay: the schedule
 * changes them between replays), else from the host values.  Writes stats_dev[0..4] = policy loss, value loss, mean entropy,
 * clip fraction, total loss (pg + vf_coef * vl - ent_coef * entropy), dlogits_dev [B][5] (the logits' type) and dvalues_dev
 * [BV] float32 = d total / d outputs.  The advantages are normalised per minibatch with the unbiased standard deviation (:577). */
int pmx_ppo_loss(const void *logits_dev, int32_t logits_bf16, const float *values_dev, const int64_t *act_dev,
                 const float *old_logp_dev, const float *adv_dev, const float *ret_dev, int32_t B, int32_t BV,
                 const float *clip_eps_dev, const float *ent_coef_dev, float clip_eps, float ent_coef, float vf_coef,
                 float *stats_dev, void *dlogits_dev, float *dvalues_dev, void *stream);

/* Minibatch assembly in one launch: for t < n (n <= 8), dst[t] row r = src[t] row (idx[t][r / m] * m + r % m), m =
 * rows_per_index[t], rows of row_bytes[t] bytes (a multiple of 4; multiples of 16 need 16-byte aligned bases), n_rows[t] output
 * rows.  The arrays of pointers and sizes are HOST arrays; the pointers in them are device pointers.  (What
 * pacman_mappo_resnet.py:560-569 does with `batch_indices` on its CPU tensors.) */
int pmx_gather_rows(int32_t n, const void *const *src_dev, void *const *dst_dev, const int64_t *const *idx_dev,
                    const int32_t *row_bytes, const int32_t *rows_per_index, const int64_t *n_rows, void *stream);
/* The same launch also writes n_values <= 8 float32 host values to consecutive device words (pmx_set_floats folded in: the scalars a
 * replayed graph reads travel with the minibatch). */
int pmx_gather_rows_set_floats(int32_t n, const void *const *src_dev, void *const *dst_dev, const int64_t *const *idx_dev,
                               const int32_t *row_bytes, const int32_t *rows_per_index, const int64_t *n_rows, float *floats_dst_dev,
                               const float *values, int32_t n_values, void *stream);
/* n <= 8 float32 host values to consecutive device words, passed as kernel arguments (the scalars a replayed hipGraph reads). */
int pmx_set_floats(float *dst_dev, const float *values, int32_t n, void *stream);

/* clip_grad_norm_(max_norm) -> Adam (no weight decay, no amsgrad) -> EMA on flat float32 buffers of n elements, two launches
 * (pacman_mappo_resnet.py:587-595).  scratch_dev: PMX_OPT_PARTIALS doubles.  lr / (1 - beta1^t) and 1 / sqrt(1 - beta2^t) are read
 * from scalars_dev[0..1] when it is not NULL (graph replay), else taken from the host arguments.  grad_dev is left clipped;
 * norm_out_dev (may be NULL) receives the gradient's 2-norm before clipping. */
#define PMX_OPT_PARTIALS 1024
int pmx_clip_adam_ema(float *grad_dev, float *param_dev, float *exp_avg_dev, float *exp_avg_sq_dev, float *ema_dev, int64_t n,
                      double *scratch_dev, const float *scalars_dev, float lr_over_bc1, float rsqrt_bc2, float beta1, float beta2,
                      float eps, float max_norm, float ema_decay, float *norm_out_dev, void *stream);
/* The same two launches with the step's loose ends folded into the second one (each pointer may be NULL): param_bf16_dev receives
 * the updated parameters rounded to bfloat16 (n elements: the copy library GEMMs read under autocast); report_sums6_dev[0..4] +=
 * reports5_dev[0..4] (the objective's five scalars, pmx_ppo_loss) and report_sums6_dev[5] += the gradient norm -- the running sums
 * a caller averages over an update.  Three small launches less at the end of a replayed 512-sample step. */
int pmx_clip_adam_ema_tail(float *grad_dev, float *param_dev, float *exp_avg_dev, float *exp_avg_sq_dev, float *ema_dev, int64_t n,
                           double *scratch_dev, const float *scalars_dev, float lr_over_bc1, float rsqrt_bc2, float beta1, float beta2,
                           float eps, float max_norm, float ema_decay, float *norm_out_dev, void *param_bf16_dev,
                           const float *reports5_dev, float *report_sums6_dev, void *stream);

/* n contiguous device tensors (float32, or bfloat16 where src_is_bf16[t]) copied / widened into dst_dev at element offsets
 * dst_offset[t], count[t] elements each, one launch per 64 tensors: the parameter gradients of a step into the flat float32
 * gradient bucket.  The four arrays are HOST arrays. */
int pmx_flatten_to_f32(int32_t n, const void *const *src_dev, const uint8_t *src_is_bf16, const int64_t *dst_offset,
                       const int32_t *count, float *dst_dev, void *stream);
/* The same with the second stage of the gradient reductions folded in: where partial_rows[t] > 0, src_dev[t] points into row 0 of a
 * float32 partial-row buffer (pmx_defer_row_sums) whose rows 1 .. partial_rows[t] lie row_stride[t] floats apart, and the
 * destination receives their sum.  partial_rows / row_stride: HOST arrays, or both NULL. */
int pmx_flatten_sum_to_f32(int32_t n, const void *const *src_dev, const uint8_t *src_is_bf16, const int32_t *partial_rows,
                           const int32_t *row_stride, const int64_t *dst_offset, const int32_t *count, float *dst_dev, void *stream);

/* ---- The actor's convolutional tower as one forward and one backward kernel ------------------------------------------
 * MAPPOAgent.actor_backbone (pacman_mappo_resnet.py:104-113 with ResidualBlock :49-67):
 *   conv3x3(8->16) GELU conv3x3(16->32) GELU 3 x [conv3x3 GroupNorm(4) GELU conv3x3 GroupNorm(4) (+x) GELU], bf16 matrix-core
 *   products with fp32 accumulation, GroupNorm and GELU in fp32 on the bf16-rounded convolution output (what bf16 autocast
 *   computes).  Boards whose padded area H*(W+2) needs 10, 11 or 28 position tiles of 16 are supported (tinyCapture,
 *   smallCapture; the 20 x 20 boards: bloxCapture and the generated mazes); pmx_actor_supported() says so and callers keep the
 *   library convolutions for the rest.
 * Parameters arrive as float32 device pointers in nn.Module order: conv_w[l] is [cout][cin][3][3], l = 0,1 the stem, then
 * conv1 / conv2 of the three blocks; gn_w / gn_b [6][32] are gn1, gn2 of the three blocks. */
typedef struct {
    const float *conv_w[8];
    const float *conv_b[8];
    const float *gn_w[6];
    const float *gn_b[6];
} pmx_actor_params;
#define PMX_ACTOR_PACK_BYTES 297984    /* bf16 operand fragments of the 8 layers (forward + input-gradient order) + fp32 biases / GroupNorm affine */
#define PMX_ACTOR_GRAD_FLOATS 74496    /* backward's fp32 gradient buffer: weight-gradient tiles + bias / GroupNorm gradients */
int pmx_actor_supported(int32_t H, int32_t W);
/* bytes of the activation save area (training forward -> backward) and of backward's scratch for B samples, and of the
 * scratch an inference-only forward needs (independent of B); any of the three pointers may be NULL */
int pmx_actor_sizes(int32_t H, int32_t W, int64_t B, int64_t *save_bytes, int64_t *scratch_bytes, int64_t *infer_scratch_bytes);
/* parameters -> pack_dev [PMX_ACTOR_PACK_BYTES]; once per optimizer step (the weights changed) or once per rollout */
int pmx_actor_pack(const pmx_actor_params *params, void *pack_dev, void *stream);
/* obs_dev [B][8][H][W] of PMX_OBS_* elements -> feat_dev [B][H*W][32] bfloat16 (channels-last; the reference's nn.Flatten order
 * is the transpose of the last two dimensions).  save_dev NULL = inference (scratch_dev then holds the blocks' skip inputs
 * between layers and is required); otherwise save_dev receives what backward needs and scratch_dev may be NULL. */
int pmx_actor_forward(const void *obs_dev, int32_t obs_dtype, const void *pack_dev, void *feat_dev, void *save_dev,
                      void *scratch_dev, int64_t B, int32_t H, int32_t W, void *stream);
/* dfeat_dev [B][H*W][32] bfloat16 -> grad_dev [PMX_ACTOR_GRAD_FLOATS] (zeroed and accumulated here, sum over the batch) */
int pmx_actor_backward(const void *obs_dev, int32_t obs_dtype, const void *pack_dev, const void *save_dev, const void *dfeat_dev,
                       void *scratch_dev, float *grad_dev, int64_t B, int32_t H, int32_t W, void *stream);
/* grad_dev -> float32 gradients in the parameters' own shapes (the pointers of `out` are written, not read) */
int pmx_actor_unpack_grads(const float *grad_dev, const pmx_actor_params *out, void *stream);

/* ---- The feed-forward half of the critic's encoder layer as one forward and one backward kernel ----------------------
 * nn.TransformerEncoderLayer(d_model 32, dim_feedforward 128, ReLU, dropout 0, norm_first False), pacman_mappo_resnet.py:138-141:
 *   y = LayerNorm(x + linear2(relu(linear1(x))))   on [tokens][32] bfloat16 rows, fp32 accumulation and LayerNorm.
 * Parameters are float32 device pointers with nn.Module shapes (linear1.weight [128][32], linear2.weight [32][128], norm2). */
/* The backward kernels of this family sum their parameter gradients in two stages (a partial row per workgroup, then a row sum;
 * float atomics from every workgroup onto the same few thousand addresses are an order of magnitude slower): grad_dev must
 * hold (1 + PMX_GRAD_PARTIAL_ROWS) rows of *_GRAD_FLOATS floats; the result is row 0. */
#define PMX_GRAD_PARTIAL_ROWS 512
#define PMX_FFN_PACK_BYTES 33664
#define PMX_FFN_GRAD_FLOATS 8416      /* dW2 [32][128], dW1 [128][32], db1 [128], db2 [32], dgamma [32], dbeta [32] */
int pmx_ffn_pack(const float *w1, const float *b1, const float *w2, const float *b2, const float *gamma, const float *beta,
                 void *pack_dev, void *stream);
int pmx_ffn_forward(const void *x_dev, const void *pack_dev, void *y_dev, int64_t tokens, float eps, void *stream);
/* recomputes the forward from x (nothing is saved): dx_dev [tokens][32] bfloat16, grad_dev [1 + PMX_GRAD_PARTIAL_ROWS][PMX_FFN_GRAD_FLOATS] (row 0 = result) */
int pmx_ffn_backward(const void *x_dev, const void *dy_dev, const void *pack_dev, void *dx_dev, float *grad_dev, int64_t tokens,
                     float eps, void *stream);

/* The other token-parallel pieces of the encoder layer (nn.MultiheadAttention's packed in-projection and its out-projection
 * followed by the residual add and norm1), same kernel scheme, [tokens][32] bfloat16 inputs:
 *   tok96:    qkv [tokens][96] = in_proj_weight [96][32] a + in_proj_bias
 *   tok32ln:  y [tokens][32]  = LayerNorm(x + out_proj.weight [32][32] a + out_proj.bias)
 * Backward returns da (and dx = the gradient of the residual input) and the parameter gradients in row 0 of grad_dev
 * [1 + PMX_GRAD_PARTIAL_ROWS][*_GRAD_FLOATS]:
 *   tok96:   dW [96][32], db [96]                  (+ 64 unused floats)
 *   tok32ln: dW [32][32], db [32], dgamma [32], dbeta [32] */
#define PMX_TOK96_PACK_BYTES 12928
#define PMX_TOK96_GRAD_FLOATS 3232
#define PMX_TOK32_PACK_BYTES 4480
#define PMX_TOK32_GRAD_FLOATS 1120
int pmx_tok96_pack(const float *w, const float *b, void *pack_dev, void *stream);
int pmx_tok96_forward(const void *a_dev, const void *pack_dev, void *y_dev, int64_t tokens, void *stream);
int pmx_tok96_backward(const void *a_dev, const void *dy_dev, const void *pack_dev, void *da_dev, float *grad_dev, int64_t tokens, void *stream);
/* pmx_tok96_backward with da = in_proj_weight^T dy + res: res_dev [tokens][32] bfloat16 (or NULL) is the gradient that reaches the same
 * tokens through the residual connection of the encoder layer (nn.TransformerEncoderLayer: norm1(x + self_attn(x)), pacman_mappo_resnet.py:
 * 138-141) -- autograd would add the two with a kernel of its own.  res_dev must not be da_dev. */
int pmx_tok96_backward_res(const void *a_dev, const void *dy_dev, const void *pack_dev, const void *res_dev, void *da_dev, float *grad_dev,
                           int64_t tokens, void *stream);
int pmx_tok32ln_pack(const float *w, const float *b, const float *gamma, const float *beta, void *pack_dev, void *stream);
int pmx_tok32ln_forward(const void *x_dev, const void *a_dev, const void *pack_dev, void *y_dev, int64_t tokens, float eps, void *stream);
int pmx_tok32ln_backward(const void *x_dev, const void *a_dev, const void *dy_dev, const void *pack_dev, void *dx_dev, void *da_dev,
                         float *grad_dev, int64_t tokens, float eps, void *stream);

/* ---- The small ends of the two heads, one kernel each way (csrc/pmx_heads.hip) ----------------------------------------
 * Actor tail (pacman_mappo_resnet.py:117-122, actor_head[1:]): logits = W2 gelu(LayerNorm_512(h)) + b2 on the 512-wide output h
 * [B][512] (bfloat16 if h_bf16, else float32) of the first head layer; logits_dev [B][5] float32; stats_dev [B][2] float32 (mean,
 * rstd; may be NULL for inference).  Backward: dh_dev [B][512] in h's type, and the parameter gradients in row 0 of grad_dev
 * [1 + PMX_HEADS_PARTIAL_ROWS][PMX_ACTOR_TAIL_GRAD_FLOATS]: dW2 [5][512], db2 [5] (+ 3 unused), dLN.weight [512], dLN.bias [512].
 * Critic tail (:143-147 critic_head on the mean over tokens, :169): value = w2 gelu(W1 mean_s tokens[b][s] + b1) + b2 with tokens
 * [B][S][32] bfloat16 (the batch-major encoder output); pooled_dev [B][32] float32 is saved for backward.  Backward: dtokens_dev
 * [B][S][32] bfloat16, scratch_dev 2 * B * 512 bfloat16, row 0 of grad_dev [1 + PMX_HEADS_PARTIAL_ROWS][PMX_CRITIC_TAIL_GRAD_FLOATS]: dW1
 * [512][32], db1 [512], dw2 [512], db2 [1] (+ 7 unused).  Parameters are float32 device pointers with nn.Module shapes; the kernels round the two linears' weights
 * and inputs to bfloat16 as autocast does and keep LayerNorm / GELU in float32. */
#define PMX_HEADS_PARTIAL_ROWS 128
#define PMX_ACTOR_TAIL_GRAD_FLOATS 3592
#define PMX_CRITIC_TAIL_GRAD_FLOATS 17416
int pmx_actor_tail_forward(const void *h_dev, int32_t h_bf16, const float *ln_w, const float *ln_b, const float *w2, const float *b2,
                           float *logits_dev, float *stats_dev, int64_t B, float eps, void *stream);
int pmx_actor_tail_backward(const void *h_dev, int32_t h_bf16, const float *stats_dev, const float *dlogits_dev, const float *ln_w,
                            const float *ln_b, const float *w2, void *dh_dev, float *grad_dev, int64_t B, void *stream);
int pmx_critic_tail_forward(const void *tokens_dev, const float *w1, const float *b1, const float *w2, const float *b2,
                            float *value_dev, float *pooled_dev, int64_t B, int32_t S, void *stream);
int pmx_critic_tail_backward(const float *pooled_dev, const float *dvalue_dev, const float *w1, const float *b1, const float *w2,
                             void *dtokens_dev, void *scratch_dev, float *grad_dev, int64_t B, int32_t S, void *stream);

/* All parameter packs of up to four encoder layers in one launch (what pmx_tok96_pack, pmx_tok32ln_pack and pmx_ffn_pack produce, into
 * the three pack buffers of each layer). */
typedef struct {
    const float *in_proj_w, *in_proj_b, *out_proj_w, *out_proj_b, *norm1_w, *norm1_b;     /* [96][32], [96], [32][32], [32], [32], [32] */
    const float *lin1_w, *lin1_b, *lin2_w, *lin2_b, *norm2_w, *norm2_b;                   /* [128][32], [128], [32][128], [32], [32], [32] */
    void *pack_in, *pack_out, *pack_ffn;    /* PMX_TOK96_PACK_BYTES, PMX_TOK32_PACK_BYTES, PMX_FFN_PACK_BYTES */
} pmx_encoder_layer_params;
int pmx_encoder_pack(int32_t n_layers, const pmx_encoder_layer_params *layers, void *stream);

/* ---- The critic's projector: conv3x3(8 -> 32) + bias + 2-D positional encoding -> batch-major tokens ---------------------
 * MAPPOAgent.critic_projector and pos_encoder (pacman_mappo_resnet.py:126-127, :69-95, :164) on boards pmx_actor_supported() accepts:
 * obs_dev [B][8][H][W] of PMX_OBS_* elements -> tokens_dev [B][H*W][32] bfloat16 (what flatten(2).permute(2, 0, 1) yields, batch-major),
 * bf16 products with fp32 accumulation, the convolution (+ bias) rounded to bfloat16 before the table (posenc_dev [H*W][32] float32) is
 * added in bfloat16, as autocast computes it.  Backward returns the weight and bias gradients only (the observations need none):
 * dw_dev [32][8][3][3], db_dev [32] float32; partial_dev is scratch of PMX_PROJ_PARTIAL_ROWS x PMX_PROJ_GRAD_ROW_FLOATS floats. */
#define PMX_PROJ_PACK_BYTES 6272
#define PMX_PROJ_PARTIAL_ROWS 256
#define PMX_PROJ_GRAD_ROW_FLOATS 2592
int pmx_proj_pack(const float *w, const float *b, void *pack_dev, void *stream);
int pmx_proj_forward(const void *obs_dev, int32_t obs_dtype, const void *pack_dev, const float *posenc_dev, void *tokens_dev, int64_t B,
                     int32_t H, int32_t W, void *stream);
int pmx_proj_backward(const void *obs_dev, int32_t obs_dtype, const void *dtokens_dev, float *partial_dev, float *dw_dev, float *db_dev,
                      int64_t B, int32_t H, int32_t W, void *stream);

/* Deferred row sums (per host thread).  The backward entry points of the pmx_ffn / pmx_tok96 / pmx_tok32ln / pmx_actor_tail / pmx_critic_tail
 * families end with a small second-stage kernel that adds their partial rows into row 0 of grad_dev.  After pmx_defer_row_sums(1) they
 * skip it; pmx_last_partial_rows() then tells how many rows the last such call left (0: row 0 is already final), and the caller adds them
 * with pmx_sum_partial_rows(grad_dev, rows, floats-per-row, stream) -- on a side stream, beside the next backward kernel. */
int pmx_defer_row_sums(int32_t on);
int pmx_last_partial_rows(void);
int pmx_sum_partial_rows(float *buf_dev, int32_t n_rows, int32_t floats, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PMX_H */
