/*
 * dcmrta_oracle.h -- TEST INFRASTRUCTURE, NOT THE PRODUCT.
 *
 * Plain-C, single-threaded, fp64 restatement of the reference's coalition-formation +
 * routing simulator (reference: env/task_env.py, worker.py:41-112).  It is the CHECKER
 * that tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg compare the HIP
 * path against.  Nothing under dcmrta_amd/ may include, link, import or execute it.
 *
 * Parity status: PINNED.  The restatement is checked (tests/test_oracle_golden.py)
 * against (i) the reference-published known answer -- CTAS-D routes replayed through
 * execute_by_route reproduce testSet_20A_50T_CONDET/metrics/metrics.csv:2 -- and
 * (ii) golden step traces produced by importing the reference in the build container
 * (tests/golden/make_golden.py, make_golden_extra.py), incl. (round 3) traces of a policy that
 * ignores the action mask (make_golden_masked.py: masked picks, surplus members, event times that
 * step backwards) and route replays of the reference with its four dynamic-arrival literals
 * substituted (make_golden_schedule.py, for orc_set_visibility).
 *
 * Every function cites the reference file:line it follows.
 */
#ifndef DCMRTA_ORACLE_H
#define DCMRTA_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_env orc_env;

/* policies understood by orc_rollout */
enum { ORC_POLICY_RANDOM = 0, ORC_POLICY_INJECTED = 1, ORC_POLICY_FIRST = 2, ORC_POLICY_NEAREST = 3, ORC_POLICY_ANY = 4 /* ignores the mask */ };

orc_env *orc_create(int A, int T);
void orc_destroy(orc_env *e);

/* env/task_env.py:57-114 (instance), :116-140 (reset + clear_decisions) */
void orc_load_instance(orc_env *e, const double *depot_xy, const double *task_xy /*[T,2]*/,
                       const int32_t *req /*[T]*/, const double *dur /*[T]*/);
void orc_clear_decisions(orc_env *e);
void orc_set_params(orc_env *e, double max_waiting_time, double max_time);

/* the choice protocol: exposed so tests can cross-check the Python mirror */
uint64_t orc_mix64(uint64_t z);
uint64_t orc_env_seed(uint64_t base, uint64_t env_index);
uint64_t orc_draw(uint64_t seed_e, uint64_t d, uint64_t slot);

/* step-wise surface mirroring the reference method names (env/task_env.py) */
int orc_next_decision(orc_env *e, int32_t *ids_out /*[A]*/, double *t_out);            /* :283-289 */
int orc_get_unique_group(orc_env *e, const int32_t *ids, int n, int32_t *group_of /*[n]*/); /* :291-298 */
void orc_task_update(orc_env *e);                                                        /* :245-281 */
void orc_agent_update(orc_env *e);                                                       /* :207-243 */
void orc_agent_step(orc_env *e, int agent, int action);                                  /* :300-324 */
void orc_mask(orc_env *e, uint8_t *mask_out /*[T+1]*/);                                  /* :192-200 + worker.py:57-61 */
void orc_agent_status(orc_env *e, int leader, float *out /*[A,6]*/);                     /* :165-180 */
void orc_task_status(orc_env *e, int leader, float *out /*[T+1,5]*/);                    /* :182-190 */
int orc_check_finished(orc_env *e);                                                      /* :366-373 */
double orc_get_now(orc_env *e);
void orc_set_now(orc_env *e, double now);
int orc_task_status_int(orc_env *e, int task);

/*
 * Full RL-mode episode: the loop of worker.py:45-87 with the keyed choice protocol.
 * policy RANDOM draws the action from slot 1; INJECTED replays inj_action[step]
 * (and, when non-NULL, inj_leader[step] / inj_nfol[step] / inj_followers[step*A..]).
 * Any of the rec_* pointers may be NULL.  Returns the number of decisions taken,
 * or -1 if cap_steps would be exceeded.
 */
int64_t orc_rollout(orc_env *e, uint64_t seed_e, uint64_t d0, int policy, int64_t cap_steps,
                    const int32_t *inj_leader, const int32_t *inj_action, const int32_t *inj_nfol,
                    const int16_t *inj_followers /*[steps,A]*/,
                    int32_t *rec_leader, int32_t *rec_action, int32_t *rec_nfol, int16_t *rec_followers,
                    double *rec_now, uint8_t *rec_mask, float *rec_agents, float *rec_tasks);

/* terminal outputs (worker.py:87,103-108; env/task_env.py:344-364,420-425) */
typedef struct {
    double reward, makespan;
    double metrics[6]; /* success_rate, makespan, time_cost, waiting_time, travel_dist, efficiency */
    int32_t truncated;
    int32_t n_finished;
} orc_summary;
void orc_summary_get(orc_env *e, orc_summary *s);
int orc_max_members_seen(orc_env *e); /* longest task['members'] list since clear_decisions (test aid) */
void orc_final_tasks(orc_env *e, uint8_t *finished, uint8_t *feasible, double *time_start, double *time_finish,
                     double *task_wait, int32_t *n_members, int32_t *n_abandoned);
void orc_final_agents(orc_env *e, double *agent_wait, double *travel_dist, uint8_t *returned, int32_t *route_len);

/* agent['route'] / agent['arrival_time'] (env/task_env.py:95-96): copies up to cap entries, returns the route length */
int orc_get_route(orc_env *e, int agent, int32_t *tasks_out, double *arrival_out, int cap);
int orc_get_members(orc_env *e, int task, int32_t *out, int cap);   /* task['members'], list order (env/task_env.py:78) */
int orc_get_abandoned(orc_env *e, int task, int32_t *out, int cap); /* task['abandoned_agent'], append order (:89) */

/* route replay: env/task_env.py:595-599 (pre_set_route), :562-593 (execute_by_route) */
void orc_pre_set_route(orc_env *e, int agent, const int32_t *actions, int n);
/* returns 0 ok, -2 if the reference would raise TypeError at :220 (pre_set_route None) */
int orc_execute_by_route(orc_env *e, int reactive);
/* dynamic-arrival schedule: visible_length = int(clip(now // period * batch + initial, initial, cap)) (:567) and the depot
 * re-arm time (next - 1) // batch * period (:221).  The reference hard-codes 20 / 20 / 10 / 100 (the defaults here). */
int orc_set_visibility(orc_env *e, int initial, int batch, int period, int cap);
void orc_finish_episode(orc_env *e); /* get_episode_reward: calculate_waiting_time + check_finished */

/* numpy add.reduce restated (pairwise summation); exposed for a unit test against numpy */
double orc_pairwise_sum(const double *a, int64_t n);

/*
 * CPU baseline: B independent envs, random policy, `episodes` consecutive episodes each
 * (auto-reset to the same instance, decision counter keeps running), one env at a time
 * per thread on `threads` pthreads.  Observations + mask are built every decision, as in
 * the reference loop.  Returns total decisions; per-env outputs are optional.
 */
int64_t orc_batch_rollout(int B, int A, int T, const double *depot /*[B,2]*/, const double *task_xy /*[B,T,2]*/,
                          const int32_t *req /*[B,T]*/, const double *dur /*[B,T]*/, const uint64_t *seeds /*[B]*/,
                          int episodes, int threads, double *reward_out /*[B] last episode*/,
                          int64_t *steps_out /*[B]*/, double *metrics_out /*[B,6] last episode*/);
/* the same, plus every episode's reward (returns_out[B, episodes]) and the last episode's finished-task count: what the
 * full-batch parity checks compare with dcm_set_return_log / dcm_summary (worker.py:87,103-108).  Any output may be NULL. */
int64_t orc_batch_rollout_ex(int B, int A, int T, const double *depot, const double *task_xy, const int32_t *req,
                             const double *dur, const uint64_t *seeds, int episodes, int threads, double *reward_out,
                             int64_t *steps_out, double *metrics_out, double *returns_out, int32_t *n_finished_out);

/*
 * Batch form of pre_set_route + execute_by_route + get_episode_reward (env/task_env.py:562-599, :420-425): B independent
 * envs on `threads` pthreads.  routes[B,A,route_cap] actions, route_len[B,A] (-1 = pre_set_route stays None);
 * visibility = the four schedule constants (NULL = the reference's).  steps_out[b] = agent_step calls of env b;
 * status_out[b] = 0 ok, 1 ended by the zero-decider / step guard, 2 the reference raises TypeError (:220).
 * Returns the total number of agent_step calls.
 */
int64_t orc_batch_replay(int B, int A, int T, const double *depot, const double *task_xy, const int32_t *req,
                         const double *dur, const int32_t *routes, const int32_t *route_len, int route_cap, int reactive,
                         const int32_t *visibility, int threads, double *reward_out, int64_t *steps_out,
                         double *metrics_out /*[B,6]*/, int32_t *n_finished_out, int32_t *status_out);

#ifdef __cplusplus
}
#endif
#endif
