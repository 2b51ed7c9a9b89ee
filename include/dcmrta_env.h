/*
 * dcmrta_env.h -- C ABI of the MI355X-native batched coalition-formation + routing env.
 *
 * The reference (marmotlab/DCMRTA) has no FFI on this path: its boundary is duck-typed
 * Python (env/task_env.py class TaskEnv, driven by worker.py:41-112).  This header is
 * the boundary a maintainer binds instead (ctypes stub in INTEGRATION.md).  Each entry
 * point cites the reference interface it replaces.
 *
 * Conventions
 *  - plain C symbols, every call returns 0 on success or a negative dcm_status;
 *    dcm_last_error() returns a thread-local message for the last failure.
 *  - every array argument is a caller-owned DEVICE pointer (tensor.data_ptr()) unless it
 *    is marked "host"; the library never frees or retains it past the call.
 *  - every call takes the hipStream_t to enqueue on (torch.cuda.current_stream().cuda_stream,
 *    passed as void*) and is asynchronous with respect to the host.
 *  - a handle is bound to one device and is not thread-safe (one host thread per GPU,
 *    like one env per Ray actor in runner.py:74).  Calls that launch work must be made with
 *    that device current (hipSetDevice / torch.cuda.device): otherwise they fail with
 *    DCM_ERR_STATE instead of launching on the wrong GPU.
 *  - there is NO CPU implementation behind this ABI: without a HIP device dcm_create fails.
 *
 * One "env step" = one leader decision in one env = one TaskEnv.step call
 * (env/task_env.py:326) plus the task_update/agent_update and observation/mask
 * construction the worker wraps around it (worker.py:57-84).
 */
#ifndef DCMRTA_ENV_H
#define DCMRTA_ENV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DCM_ABI_VERSION 5 /* v3: dcm_build_id, dcm_set_visibility, dcm_load_instances validates req on the device; v4: dcm_set_replay_placement; v5: DCM_PARAM_WIDE_MEMBERS */

typedef struct dcm_env dcm_env; /* opaque */

typedef enum {
    DCM_OK = 0,
    DCM_ERR_INVALID = -1, /* bad argument */
    DCM_ERR_HIP = -2,     /* HIP runtime error (message has hipGetErrorString) */
    DCM_ERR_NO_DEVICE = -3,
    DCM_ERR_STATE = -4    /* call not valid in the handle's current state */
} dcm_status;

/* limits of this build */
#define DCM_MAX_AGENTS 128
#define DCM_MAX_TASKS 1023
#define DCM_MAX_MEMBERS 5 /* COALITION_SIZE, parameters.py:17: members <= requirement <= 5 */
#define DCM_MAX_MEMBERS_WIDE 16 /* member slots per task of a DCM_PARAM_WIDE_MEMBERS handle */
#define DCM_FOLLOWER_COLS 4

/* per-env flag bits reported by dcm_env_status */
#define DCM_FLAG_DONE 1u       /* episode over (terminal box of worker.py:87) */
#define DCM_FLAG_FINISHED 2u   /* env.finished is True (env/task_env.py:366-373) */
#define DCM_FLAG_TRUNCATED 4u  /* zero-decider guard fired (the reference would spin) */
#define DCM_FLAG_BAD_ACTION 8u /* action outside [0, T]; with DCM_PARAM_STRICT_MASK also a task the mask forbids */
#define DCM_FLAG_OVERFLOW 16u  /* a task would exceed DCM_MAX_MEMBERS members / injected follower count too large */
#define DCM_FLAG_BAD_LEADER 32u /* injected leader/follower not in the current group */
#define DCM_FLAG_WAIT_ORDER 128u /* informational, does not freeze the env: a per-(agent, task) abandonment counter saturated \
                                  * (65535; RL mode) or, in route replay, an agent was moved to abandoned_agent lists more than 16 \
                                  * times: that agent's sum_waiting_time then adds the excess max_waiting_time terms last instead \
                                  * of in task order (env/task_env.py:363-364), i.e. it may differ in the last bits */
#define DCM_FLAG_BAD_INSTANCE 256u /* dcm_load_instances found a requirement outside 1..DCM_MAX_MEMBERS (checked on the device): \
                                    * the env never starts (DONE from dcm_reset on) until a valid instance is loaded */
#define DCM_FLAG_TYPE_ERROR 64u /* route replay: the reference raises TypeError here (env/task_env.py:220, pre_set_route None) */

/* dcm_params.flags: individual selection (Worker.run_test_IS, worker.py:159-198, skips get_unique_group) -- all agents
 * deciding at an event form ONE group and act one by one in ascending id, each alone: unless a leader / follower count is
 * injected, dcm_observe / dcm_step take the lowest pending id as the deciding agent and no followers.
 * Honoured by dcm_reset / dcm_observe / dcm_step; dcm_rollout_random always groups by location. */
#define DCM_PARAM_NO_GROUPING 1u
/* dcm_params.flags: auto-reset for the lockstep API -- when a dcm_step ends an env's episode (terminal box of worker.py:87,
 * results written to its dcm_summary row), the same call restarts the env from its loaded instance (reset + clear_decisions
 * + the first event, exactly what dcm_rollout_random does between its episodes: the decision counter keeps running) and the
 * fused observation is the first decision of the new episode; the env stays active.  dcm_env_episodes counts the finished
 * episodes.  This is SURVEY.md §8(d)'s "consecutive episodes, auto-reset to the same instance" for a policy in the loop: the
 * batch stays full instead of waiting for its longest episode.  Envs frozen by an error flag are not restarted, nor envs
 * that have finished dcm_params.auto_reset_episodes episodes (when that is non-zero).
 * The summary row of an episode that ended in an eager dcm_step may be computed LAZILY (the wave that ends an episode is the slowest
 * of its launch: it parks the final record and the reward + metrics are computed from it later, in batches): dcm_summary -- like
 * every entry point that reads or writes summary rows -- completes the waiting rows first, on the stream it is given, so callers
 * see no difference; the return log (dcm_set_return_log) and dcm_env_episodes are always up to date.  A dcm_step issued under
 * stream capture computes its terminal metrics inline and refuses (DCM_ERR_STATE) to be captured while rows of earlier eager steps
 * are still waiting: call dcm_summary or dcm_reset before the capture.  After a capture every step of the handle, eager or replayed,
 * computes its metrics inline. */
#define DCM_PARAM_AUTO_RESET 2u
/* dcm_params.flags: refuse host-supplied actions on masked tasks.  By default dcm_step simulates ANY action in [0, T] the way
 * TaskEnv.step does (env/task_env.py:326-342 never looks at the mask; worker.py:140's argmax can return a masked index): on a
 * task that is feasible or has status <= 0 the leader goes alone, the task may list more agents than it requires, and time
 * can even step backwards when an agent joins a task that is already over -- all of it restated.  The one limit is the
 * member-slot capacity: a task that would list more than DCM_MAX_MEMBERS agents freezes the env with DCM_FLAG_OVERFLOW.  With
 * this flag such an action freezes the env with DCM_FLAG_BAD_ACTION instead (the policy contract gives masked actions
 * probability 0, attention.py:74-76: a runner can use it to catch a policy that does not). */
#define DCM_PARAM_STRICT_MASK 4u
/* dcm_params.flags: DCM_MAX_MEMBERS_WIDE (16) member slots per task instead of DCM_MAX_MEMBERS (5).  The reference's member lists
 * are unbounded (env/task_env.py:321-322): a policy that ignores the mask (worker.py:140) can send more agents to a task than it
 * requires, and generate_env takes any max_coalition_size (:71).  A wide handle simulates both up to sixteen listed members per
 * task (requirements 1..16 are accepted by dcm_load_instances; a seventeenth member freezes the env with DCM_FLAG_OVERFLOW); its
 * records are 96 * T bytes larger, every shape runs the runtime-size kernels, dcm_get_members returns ids_out[B][T][16], and route
 * replay (which has its own member_cap) is not available on it.  Injected follower lists still hold at most DCM_FOLLOWER_COLS. */
#define DCM_PARAM_WIDE_MEMBERS 8u

typedef struct {
    int32_t n_envs;             /* B */
    int32_t n_agents;           /* A, identical for every env of the batch (driver.py:114-117) */
    int32_t n_tasks;            /* T */
    int32_t device;             /* HIP device ordinal */
    double max_waiting_time;    /* env/task_env.py:30 (10) */
    double max_time;            /* MAX_TIME, parameters.py:18 (100) */
    uint32_t flags;             /* DCM_PARAM_* bits */
    uint32_t auto_reset_episodes; /* with DCM_PARAM_AUTO_RESET: an env stops restarting once it has finished this many
                                   * episodes since dcm_reset (0 = it restarts forever) */
} dcm_params;

const char *dcm_last_error(void);
int dcm_abi_version(void);
/* 16 hex digits: sha256 over the kernel sources and compile flags this library was built from.  profiles/counters.json
 * stores it with every rocprof counter set, so bench.py can tell when the counters describe another binary. */
const char *dcm_build_id(void);

/* TaskEnv.__init__ (env/task_env.py:9-34): allocate SoA state for B envs of (A,T). */
int dcm_create(const dcm_params *params, dcm_env **out);
int dcm_destroy(dcm_env *env);

/* generate_env outputs (env/task_env.py:57-114) handed over as arrays:
 * depot[B,2] f64, task_xy[B,T,2] f64, req[B,T] i32 in 1..DCM_MAX_MEMBERS, dur[B,T] f64.
 * req is validated on the device: an env with a requirement outside the range gets DCM_FLAG_BAD_INSTANCE and stays frozen. */
int dcm_load_instances(dcm_env *env, const double *depot, const double *task_xy, const int32_t *req,
                       const double *dur, void *stream);

/* Ragged batch: TaskEnv(agents_range=(lo,hi), tasks_range=(lo,hi)) draws its own sizes per env
 * (env/task_env.py:58-65), which is what Runner.testing(seed) does with the default ranges
 * (runner.py:45-49, parameters.py:15-16).  Env e has n_agents[e] <= A agents and n_tasks[e] <= T tasks
 * and behaves exactly like an (n_agents[e], n_tasks[e]) env; the device arrays keep the batch shapes
 * of dcm_load_instances (rows beyond an env's own sizes are ignored).  Every output keeps the batch
 * shapes too: observation rows beyond an env's sizes are padding in the policy's convention --
 * all features -1 (attention.py:10-18, worker.py:253-257) and mask True (worker.py:258-261); getter
 * rows beyond the sizes read as 0 (next_decision NaN, current -2).  n_agents / n_tasks are HOST
 * arrays of B int32 (validated here).  dcm_load_instances switches back to a uniform batch.
 * Route replay (dcm_execute_routes) needs a uniform batch. */
int dcm_load_instances_ragged(dcm_env *env, const double *depot, const double *task_xy, const int32_t *req,
                              const double *dur, const int32_t *n_agents, const int32_t *n_tasks, void *stream);

/* reset + clear_decisions (env/task_env.py:116-140) for every env, then advance each env to its
 * first decision point (event t=0, one group of all agents; worker.py:45-51).
 * seeds[B] u64: per-env seed of the choice protocol; the decision counter restarts at 0. */
int dcm_reset(dcm_env *env, const uint64_t *seeds, void *stream);

/* What worker.py:54-68 builds before calling the policy, for all envs at once:
 *   agents_out[B,A,6] f32   get_current_agent_status  (env/task_env.py:165-180)
 *   tasks_out[B,T+1,5] f32  get_current_task_status   (env/task_env.py:182-190)
 *   mask_out[B,T+1] u8      get_unfinished_task_mask + depot bit (:192-200, worker.py:57-61); 1 = forbidden
 *   leader_out[B] i32       the deciding agent (worker.py:54), -1 for finished envs
 *   active_out[B] u8        0 once the env's episode is over
 * leader_in (nullable, [B] i32, -1 = draw): inject the leader instead of drawing it.
 * Pure function of the state: calling it twice returns the same tensors. */
int dcm_observe(dcm_env *env, float *agents_out, float *tasks_out, uint8_t *mask_out, int32_t *leader_out,
                uint8_t *active_out, const int32_t *leader_in, void *stream);

/* TaskEnv.step (env/task_env.py:326-342) for the current leader of every active env, followed by
 * task_update/agent_update (worker.py:74-76) and the advance to the next decision point:
 * next leader in the group -> next group -> check_finished (worker.py:85) -> next event
 * (worker.py:45-51) -> terminal (worker.py:87, metrics :103-108).
 *   actions[B] i32: 0 = depot, k = task k-1.
 *   leader_in / nfol_in / followers_in[B,DCM_FOLLOWER_COLS] (all nullable): injected choices;
 *   nfol_in[b] < 0 means "draw followers from the protocol"; nfol_in[b] >= 0 is honoured for the depot action too (the
 *   leader returns with exactly those followers -- agent_step(agent, 0) of individual selection -- instead of the whole group).
 * If agents_out..active_out are non-NULL the observation of the NEW decision point is written
 * in the same launch (fused observe; leader drawn from the protocol). */
int dcm_step(dcm_env *env, const int32_t *actions, const int32_t *leader_in, const int32_t *nfol_in,
             const int16_t *followers_in, float *agents_out, float *tasks_out, uint8_t *mask_out,
             int32_t *leader_out, uint8_t *active_out, void *stream);

/* Optional route history = agent['route'] / agent['arrival_time'] of the reference (env/task_env.py:95-96,314,318), the
 * input of generate_traj / generate_route (env/task_env.py:375-418, worker.py:244-251).  When set, every agent_step
 * executed by dcm_step appends (task id, -1 = depot; arrival time) to the agent's log:
 * route_task[B,A,cap] i16, route_arrival[B,A,cap] f64, route_len[B,A] i32 -- caller-owned device memory that must
 * outlive its use; all NULL disables.  dcm_reset zeroes route_len; entries beyond cap are counted but not stored.
 * (dcm_rollout_random does not log: use the lockstep API when trajectories are wanted.) */
int dcm_set_route_log(dcm_env *env, int16_t *route_task, double *route_arrival, int32_t *route_len, int32_t cap);

/* Optional log of EVERY episode's return (reward = -makespan, env/task_env.py:424; what worker.py:87 reads per episode):
 * returns[B,cap] f64, caller-owned device memory that must outlive its use; NULL / 0 disables.  The k-th episode an env
 * finishes since dcm_reset (dcm_rollout_random, dcm_step) writes returns[b, k mod cap] -- a ring, so a caller that plays
 * `cap` episodes per call finds all of them after the call (dcm_summary keeps only the last one).  Host-side setter. */
int dcm_set_return_log(dcm_env *env, double *returns, int32_t cap);

/* Config-2 hot path: every env plays `episodes` complete episodes under the uniform-random valid
 * policy inside ONE persistent launch (worker.py:45-87 with the action drawn from slot 1 of the
 * protocol); the observation tensors + mask are produced at every decision exactly as dcm_observe
 * does and written to agents_out/tasks_out/mask_out (nullable: skip the stores).  Envs restart
 * from their loaded instance between episodes; the decision counter keeps running.  An env that is in the middle of an
 * episode when the call starts first plays that episode to its end (it counts as one of the `episodes`).
 * Decision budget: max_decisions_in[B] i64 (nullable device array) or, when NULL, the scalar max_decisions -- an env
 * takes at most that many decisions in this call (< 0 = unlimited).  When the budget runs out the env stays at the
 * decision point it has reached with nothing of the pending decision applied (dcm_observe gives its observation;
 * dcm_step / a further dcm_rollout_random carry on from there with identical results), and agents_out / tasks_out /
 * mask_out hold what the kernel stored for the LAST DECISION TAKEN -- which is how the parity tests compare the
 * persistent kernel's own per-decision stores with the oracle at arbitrary decision indices.
 * steps_out[B] i64: decisions taken by each env in this call. */
int dcm_rollout_random(dcm_env *env, int32_t episodes, int64_t max_decisions, const int64_t *max_decisions_in,
                       float *agents_out, float *tasks_out, uint8_t *mask_out, int64_t *steps_out, void *stream);

/* Terminal results of the last finished episode (worker.py:87,103-108):
 * out[B,8] f64 = reward(-makespan), n_finished_tasks, success_rate, makespan, time_cost,
 *                waiting_time, travel_dist, efficiency. Rows of envs not yet done are NaN. */
int dcm_summary(dcm_env *env, double *out, void *stream);

/* Per-env flags (DCM_FLAG_*), decision counters and current time. Any pointer may be NULL. */
int dcm_env_status(dcm_env *env, uint32_t *flags_out, int64_t *decisions_out, double *now_out, void *stream);

/* episodes_out[B] i32: episodes finished by each env since dcm_reset (dcm_step with DCM_PARAM_AUTO_RESET, dcm_rollout_random). */
int dcm_env_episodes(dcm_env *env, int32_t *episodes_out, void *stream);

/* Per-task / per-agent state for parity tests and the per-env facade (any pointer may be NULL):
 * tasks: finished,feasible u8[B,T]; time_start,time_finish,sum_waiting_time f64[B,T]; status,n_members,n_abandoned i32[B,T]
 * agents: sum_waiting_time,travel_dist,next_decision,arrival f64[B,A]; x,y f64[B,A]; returned,assigned u8[B,A];
 *         current i32[B,A] (route[-1]: -2 none, -1 depot, k task); pending_group i32[B,A] (0 = not deciding in this
 *         event, g >= 1 = index of its group in get_unique_group order, env/task_env.py:291-298) */
int dcm_get_tasks(dcm_env *env, uint8_t *finished, uint8_t *feasible, double *time_start, double *time_finish,
                  double *sum_wait, int32_t *status, int32_t *n_members, int32_t *n_abandoned, void *stream);
int dcm_get_agents(dcm_env *env, double *sum_wait, double *travel_dist, double *next_decision, double *arrival,
                   double *x, double *y, uint8_t *returned, uint8_t *assigned, int32_t *current,
                   int32_t *pending_group, void *stream);

/* task['members'] (env/task_env.py:80) of every task in list order: ids_out int16[B,T,DCM_MAX_MEMBERS], -1 padded
 * (what generate_traj :390,:400 and the plotting code test membership against). */
int dcm_get_members(dcm_env *env, int16_t *ids_out, void *stream);

/* task['abandoned_agent'] (env/task_env.py:89) of every task as counts: counts_out uint16[B,A,T], [b,a,t] = number of times
 * task t has moved agent a to its abandoned_agent list in the current episode (calculate_waiting_time :358-364 only uses
 * membership and multiplicity; the order of the appends is not kept). */
int dcm_get_abandoned(dcm_env *env, uint16_t *counts_out, void *stream);

/* copy.deepcopy(env) (worker.py:33): snapshot / restore of the mutable SoA state.
 * dcm_state_bytes gives the buffer size (device memory) needed for all B envs. */
int dcm_state_bytes(dcm_env *env, size_t *bytes_out);
int dcm_clone_state(dcm_env *env, void *dst, void *stream);
int dcm_restore_state(dcm_env *env, const void *src, void *stream);

/* calculate_eulidean_distance (env/task_env.py:161-163) and travel time (:315) on the device, for the
 * known-answer test of the fp64 sqrt/divide path: dist_out[n], time_out[n] (nullable). */
int dcm_distance(const double *ax, const double *ay, const double *bx, const double *by, double *dist_out,
                 double *time_out, int64_t n, void *stream);

/* pre_set_route (env/task_env.py:595-599) for every agent of every env:
 * routes[B,A,route_cap] i32 actions in visiting order (0 = depot, k = task k-1; what baselines/CTAS-D.py:41-45 passes),
 * route_len[B,A] i32 (-1 = pre_set_route stays None, 0 = empty list).  member_cap (1..32) sizes the per-task member
 * slots of replay mode, where a task may collect more agents than it requires. Arrays are copied. */
int dcm_load_routes(dcm_env *env, const int32_t *routes, const int32_t *route_len, int32_t route_cap,
                    int32_t member_cap, void *stream);

/* The dynamic-arrival schedule of execute_by_route's reactive mode.  The reference hard-codes it:
 *   visible_length = int(np.clip(current_time // 10 * 20 + 20, 20, 100))     env/task_env.py:567
 *   depot re-arm time (next_action - 1) // 20 * 10                           env/task_env.py:221
 * i.e. 20 tasks at t = 0, 20 more every 10 time units, never more than 100 -- tasks 101.. of a larger instance never appear.
 * Here the four constants are handle parameters (initial, batch, period, cap; defaults 20, 20, 10, 100 = the reference):
 *   visible_length = int(clip(now // period * batch + initial, initial, cap)),  re-arm (next - 1) // batch * period.
 * Needs initial >= 0, batch >= 1, period >= 1, cap >= initial.  Host-side setter; takes effect at the next dcm_execute_routes. */
int dcm_set_visibility(dcm_env *env, int32_t initial, int32_t batch, int32_t period, int32_t cap);

/* Which replay kernel dcm_execute_routes runs and where it keeps its state.
 * 0 = auto (default): the register-resident kernel whenever the shape allows it -- at most 128 agents, member_cap <= 8, and at
 *     most 128 LIVE tasks, i.e. tasks that can ever get a member: all of them without dynamic arrivals, tasks 1..cap with them
 *     (an agent is never sent to a task that is not visible yet, env/task_env.py:578-584, and visible <= cap, :567).  BASELINE
 *     config 5 (100A/500T at the reference's cap of 100) and every 20A/50T-class replay qualify.  Otherwise the general kernel,
 *     with its "replay scratch" (member arrival times, time_finish, wake-up times, travel distance, latest arrival; allocated by
 *     dcm_load_routes: 8 T (member_cap + 1) + 16 A + 4 T bytes per env) in LDS when the batch has at most four envs per CU --
 *     every env is resident at once anyway and one wave's latency is what counts -- else in HBM.
 * 1 / 2: always the general kernel, replay scratch in LDS / in HBM (for tests and measurements).
 * Results do not depend on it. */
int dcm_set_replay_placement(dcm_env *env, int32_t placement);

/* execute_by_route (env/task_env.py:562-593; max_waiting_time 100, cut-off 200) followed by get_episode_reward
 * (:420-425), whole episode in one launch.  reactive != 0 enables dynamic task visibility (:566-567,:578-584 and the
 * depot branch of agent_update :213-224).  Results: dcm_summary rows + the optional arrays
 * steps_out i64[B] (agent_step calls), flags_out u32[B] (DCM_FLAG_*), finished u8[B,T], time_start/time_finish/
 * task_wait f64[B,T], n_members i32[B,T], agent_wait/travel_dist f64[B,A], returned u8[B,A]. */
int dcm_execute_routes(dcm_env *env, int32_t reactive, int64_t *steps_out, uint32_t *flags_out, uint8_t *finished,
                       double *time_start, double *time_finish, double *task_wait, int32_t *n_members,
                       double *agent_wait, double *travel_dist, uint8_t *returned, void *stream);

/* bytes of canonical state per env: S = 64 + 48*A + 96*T.  (Handles with A <= 20 and T <= 50 keep their
 * records in the fixed Lay{20,50} layout -- 5824 B per env -- so that every shape of the reference's training range runs
 * the same constant-offset kernels; dcm_state_bytes reports what is really allocated.) */
int dcm_record_bytes(dcm_env *env, size_t *bytes_out);

#ifdef __cplusplus
}
#endif
#endif /* DCMRTA_ENV_H */
