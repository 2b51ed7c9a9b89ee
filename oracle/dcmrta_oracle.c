/*
 * dcmrta_oracle.c -- TEST INFRASTRUCTURE, NOT THE PRODUCT (see dcmrta_oracle.h).
 *
 * Literal, sequential fp64 restatement of the reference simulator.  Unlike the HIP
 * path (compact SoA state, one wavefront per env) this file keeps the reference's own
 * data model -- per-agent route / arrival_time lists, per-task ordered member lists,
 * abandoned lists -- so that the equivalence "compact state == list state" is itself
 * something the parity tests verify.
 *
 * Build:  gcc -O2 -ffp-contract=off -fPIC -shared -pthread  (see oracle/Makefile).
 * -ffp-contract=off is REQUIRED: the only fused multiply-add is the explicit fma() in
 * orc_dist(), which restates what numpy's norm does on the reference machine.
 */
#include "dcmrta_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ tiny python-list stand-ins */
typedef struct { int *v; int n, cap; } ivec;
typedef struct { double *v; int n, cap; } dvec;

static void iv_push(ivec *l, int x) {
    if (l->n == l->cap) { l->cap = l->cap ? 2 * l->cap : 8; l->v = (int *)realloc(l->v, sizeof(int) * l->cap); }
    l->v[l->n++] = x;
}
static void dv_push(dvec *l, double x) {
    if (l->n == l->cap) { l->cap = l->cap ? 2 * l->cap : 8; l->v = (double *)realloc(l->v, sizeof(double) * l->cap); }
    l->v[l->n++] = x;
}
static int iv_index(const ivec *l, int x) { for (int i = 0; i < l->n; i++) if (l->v[i] == x) return i; return -1; }
static void iv_remove_value(ivec *l, int x) { /* list.remove(x): first occurrence */
    int i = iv_index(l, x);
    if (i < 0) { fprintf(stderr, "oracle: list.remove(x): x not in list\n"); abort(); }
    memmove(l->v + i, l->v + i + 1, sizeof(int) * (l->n - i - 1));
    l->n--;
}

struct orc_env {
    int A, T;
    double mwt;      /* max_waiting_time, env/task_env.py:30 */
    double max_time; /* MAX_TIME, parameters.py:18 */
    double now;      /* current_time :28 */
    int finished;    /* :31 */
    int reactive, visible_length; /* :33-34 */
    /* dynamic-arrival schedule of execute_by_route: the reference hard-codes 20 / 20 / 10 / 100 (:567, :221); kept as
     * parameters so that a generalised schedule (e.g. all 500 tasks of a 100A/500T instance eventually visible) can be
     * replayed -- a literal substitution of those four constants, pinned by tests/golden/make_golden_schedule.py */
    int vis_initial, vis_batch, vis_period, vis_cap;
    int truncated;   /* guard flag, not in the reference (SURVEY §5 hazard) */
    int max_members_seen; /* test aid: longest task['members'] list of the episode (the HIP env holds 5 per task in RL mode) */
    int type_error;  /* the reference would have raised TypeError (:220) */
    double depot[2];
    ivec depot_members; /* :112 */
    /* task_dic :76-89 */
    double *tx, *ty, *tdur, *ts, *tf, *task_wait;
    int *req, *status, *feasible, *tfin;
    ivec *members, *abandoned;
    /* agent_dic :91-110 */
    double *ax, *ay, *nd, *tdist, *agent_wait;
    ivec *route;
    dvec *arrival;
    int *returned, *assigned;
    double *scr_d; int *scr_i; int32_t *scr_ids; /* scratch, so the hot loop never calls malloc */
    ivec *preset;     /* pre_set_route; consumed from the front via preset_head */
    int *preset_head;
    int *preset_none; /* pre_set_route is None */
};

/* ------------------------------------------------------------------ choice protocol */
#define ORC_GAMMA 0x9E3779B97F4A7C15ULL
uint64_t orc_mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
uint64_t orc_env_seed(uint64_t base, uint64_t e) { return orc_mix64(base + ORC_GAMMA * (e + 1)); }
/* 32-bit word `slot` of the decision's stream hi(key_1), lo(key_1), hi(key_2), lo(key_2), ... */
uint64_t orc_draw(uint64_t seed_e, uint64_t d, uint64_t slot) {
    uint64_t key = orc_mix64(seed_e + ORC_GAMMA * (d + 1));
    for (uint64_t i = 0; i < slot / 2; i++) key = orc_mix64(key + ORC_GAMMA);
    return (slot % 2 == 0) ? (key >> 32) : (key & 0xFFFFFFFFULL);
}
/* multiply-high range reduction: r in [0,2^32) -> [0,n) */
static int orc_below(uint64_t r, int n) { return (int)((r * (uint64_t)n) >> 32); }

/* ------------------------------------------------------------------ numpy add.reduce (pairwise) */
double orc_pairwise_sum(const double *a, int64_t n) {
    if (n < 8) {
        double res = 0.;
        for (int64_t i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8], res;
        int64_t i;
        for (i = 0; i < 8; i++) r[i] = a[i];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        return orc_pairwise_sum(a, n2) + orc_pairwise_sum(a + n2, n - n2);
    }
}

/* Python float floor division (floatobject.c float_floor_div / numpy npy_divmod) */
static double py_floordiv(double vx, double wx) {
    double mod = fmod(vx, wx), div = (vx - mod) / wx, fl;
    if (mod != 0.0) { if ((wx < 0) != (mod < 0)) { mod += wx; div -= 1.0; } }
    if (div != 0.0) { fl = floor(div); if (div - fl > 0.5) fl += 1.0; } else fl = copysign(0.0, vx / wx);
    return fl;
}

/* ------------------------------------------------------------------ lifecycle */
orc_env *orc_create(int A, int T) {
    orc_env *e = (orc_env *)calloc(1, sizeof(orc_env));
    e->A = A; e->T = T; e->mwt = 10.0; e->max_time = 100.0;
    e->vis_initial = 20; e->vis_batch = 20; e->vis_period = 10; e->vis_cap = 100;   /* env/task_env.py:567, :221 */
#define DA(p, n) e->p = calloc((size_t)(n), sizeof(*e->p))
    DA(tx, T); DA(ty, T); DA(tdur, T); DA(ts, T); DA(tf, T); DA(task_wait, T);
    DA(req, T); DA(status, T); DA(feasible, T); DA(tfin, T); DA(members, T); DA(abandoned, T);
    DA(ax, A); DA(ay, A); DA(nd, A); DA(tdist, A); DA(agent_wait, A); DA(route, A); DA(arrival, A);
    DA(returned, A); DA(assigned, A); DA(preset, A); DA(preset_head, A); DA(preset_none, A);
    DA(scr_d, 2 * (A + 1)); DA(scr_i, A + 1); DA(scr_ids, A + 1);
#undef DA
    return e;
}
void orc_destroy(orc_env *e) {
    if (!e) return;
    for (int t = 0; t < e->T; t++) { free(e->members[t].v); free(e->abandoned[t].v); }
    for (int a = 0; a < e->A; a++) { free(e->route[a].v); free(e->arrival[a].v); free(e->preset[a].v); }
    free(e->depot_members.v);
    free(e->tx); free(e->ty); free(e->tdur); free(e->ts); free(e->tf); free(e->task_wait);
    free(e->req); free(e->status); free(e->feasible); free(e->tfin); free(e->members); free(e->abandoned);
    free(e->ax); free(e->ay); free(e->nd); free(e->tdist); free(e->agent_wait); free(e->route); free(e->arrival);
    free(e->returned); free(e->assigned); free(e->preset); free(e->preset_head); free(e->preset_none);
    free(e->scr_d); free(e->scr_i); free(e->scr_ids);
    free(e);
}
void orc_set_params(orc_env *e, double mwt, double max_time) { e->mwt = mwt; e->max_time = max_time; }

/* env/task_env.py:129-140 clear_decisions (+ reset :116-127: time 0, not finished) */
void orc_clear_decisions(orc_env *e) {
    for (int t = 0; t < e->T; t++) {
        e->members[t].n = 0; e->abandoned[t].n = 0;
        e->tfin[t] = 0; e->status[t] = e->req[t]; e->feasible[t] = 0;
        e->ts[t] = 0.0; e->tf[t] = 0.0; e->task_wait[t] = 0.0;
    }
    for (int a = 0; a < e->A; a++) {
        e->route[a].n = 0; e->arrival[a].n = 0;
        e->ax[a] = e->depot[0]; e->ay[a] = e->depot[1];
        e->nd[a] = 0.0; e->tdist[a] = 0.0; e->assigned[a] = 0; e->agent_wait[a] = 0.0; e->returned[a] = 0;
        e->preset[a].n = 0; e->preset_head[a] = 0; e->preset_none[a] = 1;
    }
    e->depot_members.n = 0;
    e->now = 0.0; e->finished = 0; e->truncated = 0; e->type_error = 0; e->max_members_seen = 0;
    e->reactive = 0; e->visible_length = 0;
}

/* env/task_env.py:57-114: tasks (location, requirements, time), agents all at the depot */
void orc_load_instance(orc_env *e, const double *depot_xy, const double *task_xy, const int32_t *req, const double *dur) {
    e->depot[0] = depot_xy[0]; e->depot[1] = depot_xy[1];
    for (int t = 0; t < e->T; t++) { e->tx[t] = task_xy[2 * t]; e->ty[t] = task_xy[2 * t + 1]; e->req[t] = req[t]; e->tdur[t] = dur[t]; }
    orc_clear_decisions(e);
}

double orc_get_now(orc_env *e) { return e->now; }
void orc_set_now(orc_env *e, double now) { e->now = now; }
int orc_task_status_int(orc_env *e, int t) { return e->status[t]; }

/* ------------------------------------------------------------------ primitives */
/* env/task_env.py:161-163: np.linalg.norm(a - b) on a 2-vector == sqrt(fma(dy,dy,dx*dx))
 * on the reference machine (OpenBLAS ddot; pinned by tests/golden/distance_kat.npz). */
static double orc_dist(double ax, double ay, double bx, double by) {
    double dx = ax - bx, dy = ay - by;
    return sqrt(fma(dy, dy, dx * dx));
}

/* env/task_env.py:202-205: arrival at the LAST occurrence of task_id in the agent's route */
static double get_arrival_time(orc_env *e, int agent, int task_id) {
    const ivec *r = &e->route[agent];
    for (int i = r->n - 1; i >= 0; i--) if (r->v[i] == task_id) return e->arrival[agent].v[i];
    fprintf(stderr, "oracle: IndexError in get_arrival_time(agent=%d, task=%d)\n", agent, task_id);
    abort();
}

static int all_feasible(orc_env *e, int upto) {
    if (upto > e->T) upto = e->T;
    for (int t = 0; t < upto; t++) if (!e->feasible[t]) return 0;
    return 1;
}

/* env/task_env.py:245-281 */
void orc_task_update(orc_env *e) {
    double *arrival = e->scr_d;
    int *drop = e->scr_i;
    for (int t = 0; t < e->T; t++) {
        if (!e->feasible[t]) {                                             /* :249 */
            ivec *mem = &e->members[t];
            int n = mem->n;                                                /* :250 */
            for (int j = 0; j < n; j++) arrival[j] = get_arrival_time(e, mem->v[j], t); /* :251 */
            e->status[t] = e->req[t] - n;                                  /* :252 */
            if (e->status[t] <= 0) {                                       /* :254 */
                double mx = arrival[0], mn = arrival[0];
                for (int j = 1; j < n; j++) { if (arrival[j] > mx) mx = arrival[j]; if (arrival[j] < mn) mn = arrival[j]; }
                if (mx - mn <= e->mwt) {                                   /* :255 */
                    e->ts[t] = mx;                                         /* :256 */
                    e->tf[t] = mx + e->tdur[t];                            /* :257 */
                    e->feasible[t] = 1;                                    /* :258 */
                } else {
                    e->feasible[t] = 0;                                    /* :261 */
                    double thr = mx - e->mwt;
                    int nd = 0;
                    for (int j = 0; j < n; j++) if (arrival[j] <= thr) drop[nd++] = mem->v[j]; /* :262 */
                    for (int j = 0; j < nd; j++) { iv_remove_value(mem, drop[j]); iv_push(&e->abandoned[t], drop[j]); } /* :263-265 */
                }
            } else {
                e->feasible[t] = 0;                                        /* :267 */
                /* :268-271 -- iterating the list while removing from it: the python list
                 * iterator advances its index after every body execution, so the element
                 * that slides into slot i after a removal is skipped (quirk Q1). */
                for (int i = 0; i < mem->n; i++) {
                    int m = mem->v[i];
                    if (e->now - get_arrival_time(e, m, t) >= e->mwt) {    /* :269 */
                        iv_remove_value(mem, m);                           /* :270 */
                        iv_push(&e->abandoned[t], m);                      /* :271 */
                    }
                }
            }
        } else {
            if (e->now >= e->tf[t]) e->tfin[t] = 1;                        /* :273-274 */
        }
    }
    /* depot :277-280 */
    int allf = all_feasible(e, e->T);
    for (int i = 0; i < e->depot_members.n; i++) {
        int m = e->depot_members.v[i];
        if (e->now >= get_arrival_time(e, m, -1) && allf) e->returned[m] = 1;
    }
}

/* env/task_env.py:207-243 */
void orc_agent_update(orc_env *e) {
    for (int a = 0; a < e->A; a++) {
        if (e->arrival[a].n > 0) {                                          /* :209 */
            int last = e->route[a].v[e->route[a].n - 1];
            if (last == -1) {                                               /* :212 */
                if (e->reactive) {                                          /* :213 */
                    if (all_feasible(e, e->visible_length)) {               /* :214 */
                        e->nd[a] = NAN;                                     /* :215 */
                    } else {
                        int remaining = e->preset[a].n - e->preset_head[a];
                        if (!e->preset_none[a] && remaining == 0) {         /* :217 */
                            e->nd[a] = NAN;                                 /* :218 */
                        } else {
                            if (e->preset_none[a]) { e->type_error = 1; return; } /* :220 raises TypeError */
                            int next_action = e->preset[a].v[e->preset_head[a]];  /* :220 */
                            /* :221 python int floor division */
                            int q = (next_action - 1) / e->vis_batch; if ((next_action - 1) % e->vis_batch != 0 && (next_action - 1) < 0) q--;
                            double ndt = (double)(q * e->vis_period);
                            double v = get_arrival_time(e, a, -1);          /* :222 np.max([..]) */
                            if (ndt > v) v = ndt;
                            if (e->now > v) v = e->now;
                            e->nd[a] = v;
                            if (iv_index(&e->depot_members, a) >= 0) iv_remove_value(&e->depot_members, a); /* :223-224 */
                        }
                    }
                } else {
                    e->nd[a] = NAN;                                         /* :226 */
                }
            } else {
                int K = last;                                               /* :228 */
                if (e->feasible[K]) {                                       /* :229 */
                    if (iv_index(&e->members[K], a) >= 0) {                 /* :230 */
                        e->nd[a] = e->tf[K];                                /* :231 */
                        if (e->now >= e->ts[K]) e->assigned[a] = 1;         /* :232-233 */
                    } else {
                        e->nd[a] = get_arrival_time(e, a, K) + e->mwt;      /* :235 */
                        e->assigned[a] = 0;                                 /* :236 */
                    }
                } else {
                    e->nd[a] = get_arrival_time(e, a, K) + e->mwt;          /* :238-239 */
                    e->assigned[a] = 0;                                     /* :240 */
                }
            }
        }
    }
}

/* env/task_env.py:283-289; returns the number of deciding agents */
int orc_next_decision(orc_env *e, int32_t *ids, double *t_out) {
    int any = 0;
    double tmin = 0.0;
    for (int a = 0; a < e->A; a++) {
        if (isnan(e->nd[a])) continue;
        if (!any || e->nd[a] < tmin) tmin = e->nd[a];
        any = 1;
    }
    if (!any) {                                                             /* :285-286 */
        double mx = 0.0;                                                    /* max(...) over "max(x) if x else 0" */
        int first = 1;
        for (int a = 0; a < e->A; a++) {
            double v = 0.0;
            if (e->arrival[a].n) { v = e->arrival[a].v[0]; for (int i = 1; i < e->arrival[a].n; i++) if (e->arrival[a].v[i] > v) v = e->arrival[a].v[i]; }
            if (first || v > mx) mx = v;
            first = 0;
        }
        *t_out = mx;
        return 0;
    }
    int n = 0;
    for (int a = 0; a < e->A; a++) if (e->nd[a] == tmin) ids[n++] = a;       /* :288 exact equality */
    *t_out = tmin;                                                          /* :287 */
    return n;
}

/* env/task_env.py:291-298: groups = rows of np.unique(location, axis=0) -> ascending (x, then y);
 * group_of[i] = index of the group ids[i] belongs to; returns the number of groups. */
int orc_get_unique_group(orc_env *e, const int32_t *ids, int n, int32_t *group_of) {
    double *ux = (double *)malloc(sizeof(double) * (n + 1)), *uy = (double *)malloc(sizeof(double) * (n + 1));
    int ng = 0;
    for (int i = 0; i < n; i++) {
        double x = e->ax[ids[i]], y = e->ay[ids[i]];
        int found = 0;
        for (int g = 0; g < ng; g++) if (ux[g] == x && uy[g] == y) { found = 1; break; }
        if (!found) { ux[ng] = x; uy[ng] = y; ng++; }
    }
    /* lexicographic insertion sort of the unique rows */
    for (int i = 1; i < ng; i++) {
        double x = ux[i], y = uy[i];
        int j = i - 1;
        while (j >= 0 && (ux[j] > x || (ux[j] == x && uy[j] > y))) { ux[j + 1] = ux[j]; uy[j + 1] = uy[j]; j--; }
        ux[j + 1] = x; uy[j + 1] = y;
    }
    for (int i = 0; i < n; i++) {
        double x = e->ax[ids[i]], y = e->ay[ids[i]];
        for (int g = 0; g < ng; g++) if (ux[g] == x && uy[g] == y) { group_of[i] = g; break; }
    }
    free(ux); free(uy);
    return ng;
}

/* env/task_env.py:300-324 */
void orc_agent_step(orc_env *e, int agent, int action) {
    int task_id = action - 1;                                               /* :307 */
    double tx, ty;
    ivec *members;
    if (task_id != -1) { tx = e->tx[task_id]; ty = e->ty[task_id]; members = &e->members[task_id]; }
    else { tx = e->depot[0]; ty = e->depot[1]; members = &e->depot_members; }
    iv_push(&e->route[agent], task_id);                                     /* :314 */
    double d = orc_dist(e->ax[agent], e->ay[agent], tx, ty);
    double travel_time = d / 0.2;                                           /* :315 velocity :99 */
    e->tdist[agent] += d;                                                   /* :317 */
    dv_push(&e->arrival[agent], e->now + travel_time);                      /* :318 */
    e->ax[agent] = tx; e->ay[agent] = ty;                                   /* :320 */
    if (iv_index(members, agent) < 0) iv_push(members, agent);              /* :321-322 */
    if (task_id != -1 && members->n > e->max_members_seen) e->max_members_seen = members->n;
}
int orc_max_members_seen(orc_env *e) { return e->max_members_seen; }

/* env/task_env.py:192-200 + depot bit worker.py:57-61 (1 = forbidden) */
void orc_mask(orc_env *e, uint8_t *mask) {
    int all = 1;
    for (int t = 0; t < e->T; t++) {
        int unfinished = (!e->feasible[t]) && (e->status[t] > 0);           /* :199 */
        mask[t + 1] = (uint8_t)!unfinished;                                 /* :193 */
        if (unfinished) all = 0;
    }
    mask[0] = all ? 0 : 1;                                                  /* worker.py:58-61 */
}

/* env/task_env.py:165-180, cast to float32 as worker.py:62 */
void orc_agent_status(orc_env *e, int leader, float *out) {
    for (int a = 0; a < e->A; a++) {
        double travel = 0.0, waiting = 0.0, remaining = 0.0;
        int n = e->route[a].n;
        if (n > 0 && e->route[a].v[n - 1] >= 0) {                           /* :168 */
            int K = e->route[a].v[n - 1];
            double arr = get_arrival_time(e, a, K);
            double x = arr - e->now; travel = x > 0.0 ? x : 0.0;            /* :169 */
            if (e->now <= e->ts[K]) { double w = e->now - arr; waiting = w > 0.0 ? w : 0.0; } /* :170 */
            if (e->now >= e->ts[K]) { double r = e->ts[K] + e->tdur[K] - e->now; remaining = r > 0.0 ? r : 0.0; } /* :171 */
        }
        float *row = out + 6 * a;                                           /* :176-177 */
        row[0] = (float)travel; row[1] = (float)remaining; row[2] = (float)waiting;
        row[3] = (float)(e->ax[leader] - e->ax[a]); row[4] = (float)(e->ay[leader] - e->ay[a]);
        row[5] = (float)e->assigned[a];
    }
}

/* env/task_env.py:182-190, cast to float32 as worker.py:64 */
void orc_task_status(orc_env *e, int leader, float *out) {
    out[0] = 0.f; out[1] = 0.f; out[2] = 0.f;                               /* :188 */
    out[3] = (float)(e->depot[0] - e->ax[leader]); out[4] = (float)(e->depot[1] - e->ay[leader]);
    for (int t = 0; t < e->T; t++) {                                        /* :185-186 */
        float *row = out + 5 * (t + 1);
        row[0] = (float)e->status[t]; row[1] = (float)e->req[t]; row[2] = (float)e->tdur[t];
        row[3] = (float)(e->tx[t] - e->ax[leader]); row[4] = (float)(e->ty[t] - e->ay[leader]);
    }
}

/* env/task_env.py:366-373 */
int orc_check_finished(orc_env *e) {
    int32_t *ids = e->scr_ids;
    double t;
    int n = orc_next_decision(e, ids, &t), fin = 0;
    if (n == 0) {
        e->now = t;                                                         /* :369 */
        fin = 1;
        for (int a = 0; a < e->A; a++) if (!e->returned[a]) fin = 0;        /* :370 */
        for (int k = 0; k < e->T; k++) if (!e->tfin[k]) fin = 0;
    }
    return fin;
}

/* env/task_env.py:344-364 */
static void calculate_waiting_time(orc_env *e) {
    double *arrival = e->scr_d, *tmp = e->scr_d + (e->A + 1);
    for (int a = 0; a < e->A; a++) e->agent_wait[a] = 0.0;                  /* :345-346 */
    for (int t = 0; t < e->T; t++) {
        ivec *mem = &e->members[t];
        int n = mem->n;
        double mx = 0.0;
        for (int j = 0; j < n; j++) { arrival[j] = get_arrival_time(e, mem->v[j], t); if (j == 0 || arrival[j] > mx) mx = arrival[j]; } /* :348 */
        double ab = (double)e->abandoned[t].n * e->mwt;
        if (n != 0) {
            if (e->feasible[t]) { for (int j = 0; j < n; j++) tmp[j] = mx - arrival[j]; }       /* :351 */
            else { for (int j = 0; j < n; j++) tmp[j] = e->now - arrival[j]; }                  /* :354 */
            e->task_wait[t] = orc_pairwise_sum(tmp, n) + ab;
        } else {
            e->task_wait[t] = ab;                                           /* :357 */
        }
        for (int j = 0; j < n; j++) {                                       /* :358-362 */
            int m = mem->v[j];
            if (e->feasible[t]) e->agent_wait[m] += mx - get_arrival_time(e, m, t);
            else { double w = e->now - get_arrival_time(e, m, t); e->agent_wait[m] += (w > 0.0) ? w : 0.0; }
        }
        for (int j = 0; j < e->abandoned[t].n; j++) e->agent_wait[e->abandoned[t].v[j]] += e->mwt; /* :363-364 */
    }
}

/* env/task_env.py:420-425 get_episode_reward (reward = -now is read via orc_summary_get) */
void orc_finish_episode(orc_env *e) {
    calculate_waiting_time(e);   /* :421 */
    (void)orc_check_finished(e); /* :422 (may move `now` to the last arrival; result unused) */
}

void orc_summary_get(orc_env *e, orc_summary *s) {
    int nf = 0;
    for (int t = 0; t < e->T; t++) nf += e->tfin[t];
    s->reward = -e->now;                                                    /* :424 */
    s->makespan = e->now;
    s->truncated = e->truncated;
    s->n_finished = nf;
    s->metrics[0] = (double)nf / (double)e->T;                              /* worker.py:103 */
    s->metrics[1] = e->now;                                                 /* :104 */
    s->metrics[2] = orc_pairwise_sum(e->ts, e->T) / (double)e->T;           /* :105 nanmean(time_start) */
    s->metrics[3] = orc_pairwise_sum(e->agent_wait, e->A) / (double)e->A;   /* :106 */
    s->metrics[4] = orc_pairwise_sum(e->tdist, e->A);                       /* :107 */
    s->metrics[5] = orc_pairwise_sum(e->task_wait, e->T) / (double)e->T;    /* :108 */
}

void orc_final_tasks(orc_env *e, uint8_t *finished, uint8_t *feasible, double *time_start, double *time_finish,
                     double *task_wait, int32_t *n_members, int32_t *n_abandoned) {
    for (int t = 0; t < e->T; t++) {
        if (finished) finished[t] = (uint8_t)e->tfin[t];
        if (feasible) feasible[t] = (uint8_t)e->feasible[t];
        if (time_start) time_start[t] = e->ts[t];
        if (time_finish) time_finish[t] = e->tf[t];
        if (task_wait) task_wait[t] = e->task_wait[t];
        if (n_members) n_members[t] = e->members[t].n;
        if (n_abandoned) n_abandoned[t] = e->abandoned[t].n;
    }
}
void orc_final_agents(orc_env *e, double *agent_wait, double *travel_dist, uint8_t *returned, int32_t *route_len) {
    for (int a = 0; a < e->A; a++) {
        if (agent_wait) agent_wait[a] = e->agent_wait[a];
        if (travel_dist) travel_dist[a] = e->tdist[a];
        if (returned) returned[a] = (uint8_t)e->returned[a];
        if (route_len) route_len[a] = e->route[a].n;
    }
}

int orc_get_route(orc_env *e, int agent, int32_t *tasks_out, double *arrival_out, int cap) {
    int n = e->route[agent].n;
    for (int i = 0; i < n && i < cap; i++) { tasks_out[i] = e->route[agent].v[i]; arrival_out[i] = e->arrival[agent].v[i]; }
    return n;
}

/* task['members'] / task['abandoned_agent'] lists (env/task_env.py:78,89) in list order */
int orc_get_members(orc_env *e, int task, int32_t *out, int cap) {
    int n = e->members[task].n;
    for (int i = 0; i < n && i < cap; i++) out[i] = e->members[task].v[i];
    return n;
}
int orc_get_abandoned(orc_env *e, int task, int32_t *out, int cap) {
    int n = e->abandoned[task].n;
    for (int i = 0; i < n && i < cap; i++) out[i] = e->abandoned[task].v[i];
    return n;
}

/* ------------------------------------------------------------------ RL-mode episode (worker.py:45-87) */
static int policy_pick(orc_env *e, int policy, const uint8_t *mask, int leader, uint64_t seed_e, uint64_t d) {
    int T1 = e->T + 1;
    if (policy == ORC_POLICY_RANDOM) {
        int nv = 0;
        for (int k = 0; k < T1; k++) nv += !mask[k];
        int idx = orc_below(orc_draw(seed_e, d, 1), nv);
        for (int k = 0; k < T1; k++) if (!mask[k]) { if (idx == 0) return k; idx--; }
    } else if (policy == ORC_POLICY_ANY) {
        /* a policy that does not respect the mask (worker.py:140's argmax can return a masked index; TaskEnv.step has no
         * check, env/task_env.py:326-342): 1 draw in 16 the depot whatever the mask says, 1 in 4 ANY task, else a valid action */
        uint32_t r = (uint32_t)orc_draw(seed_e, d, 1);
        if (r % 16u == 1u) return 0;
        if (r % 4u == 0u) return 1 + (int)((r >> 4) % (uint32_t)e->T);
        int nv = 0;
        for (int k = 0; k < T1; k++) nv += !mask[k];
        int idx = orc_below(r, nv);
        for (int k = 0; k < T1; k++) if (!mask[k]) { if (idx == 0) return k; idx--; }
    } else if (policy == ORC_POLICY_FIRST) {
        for (int k = 0; k < T1; k++) if (!mask[k]) return k;
    } else if (policy == ORC_POLICY_NEAREST) {
        if (!mask[0]) return 0;
        int best = -1; double bd = 0.0;
        for (int k = 1; k < T1; k++) if (!mask[k]) {
            double dd = orc_dist(e->ax[leader], e->ay[leader], e->tx[k - 1], e->ty[k - 1]);
            if (best < 0 || dd < bd) { best = k; bd = dd; }
        }
        return best;
    }
    return 0;
}

int64_t orc_rollout(orc_env *e, uint64_t seed_e, uint64_t d0, int policy, int64_t cap_steps,
                    const int32_t *inj_leader, const int32_t *inj_action, const int32_t *inj_nfol,
                    const int16_t *inj_followers,
                    int32_t *rec_leader, int32_t *rec_action, int32_t *rec_nfol, int16_t *rec_followers,
                    double *rec_now, uint8_t *rec_mask, float *rec_agents, float *rec_tasks) {
    const int A = e->A, T = e->T;
    int32_t *ids = (int32_t *)malloc(sizeof(int32_t) * A), *gof = (int32_t *)malloc(sizeof(int32_t) * A);
    int *group = (int *)malloc(sizeof(int) * A), *members = (int *)malloc(sizeof(int) * A);
    uint8_t *mask = (uint8_t *)malloc((size_t)T + 1);
    float *ag = (float *)malloc(sizeof(float) * 6 * A), *tk = (float *)malloc(sizeof(float) * 5 * (T + 1));
    int64_t step = 0;
    uint64_t d = d0;
    int empty_passes = 0;
    while (!e->finished && e->now < e->max_time) {                          /* worker.py:45 */
        double t;
        int n = orc_next_decision(e, ids, &t);                              /* :47 */
        int ng = n ? orc_get_unique_group(e, ids, n, gof) : 0;              /* :48 */
        e->now = t;                                                         /* :49 */
        orc_task_update(e);                                                 /* :50 */
        orc_agent_update(e);                                                /* :51 */
        if (ng == 0) { if (++empty_passes > 4) { e->truncated = 1; break; } } else empty_passes = 0; /* guard */
        for (int g = 0; g < ng; g++) {                                      /* :52 */
            int glen = 0;
            for (int i = 0; i < n; i++) if (gof[i] == g) group[glen++] = ids[i];
            while (glen > 0) {                                              /* :53 */
                if (step >= cap_steps) { step = -1; goto done; }
                int leader = inj_leader ? inj_leader[step] : group[orc_below(orc_draw(seed_e, d, 0), glen)]; /* :54 */
                orc_mask(e, mask);                                          /* :57-61 */
                orc_agent_status(e, leader, ag);                            /* :62 */
                orc_task_status(e, leader, tk);                             /* :64 */
                int action = (policy == ORC_POLICY_INJECTED) ? inj_action[step] : policy_pick(e, policy, mask, leader, seed_e, d);
                /* env.step: env/task_env.py:326-342 */
                int vacancy = (action - 1 >= 0 && action - 1 < T) ? e->status[action - 1] : glen; /* :327 */
                int li = -1;
                for (int i = 0; i < glen; i++) if (group[i] == leader) li = i;
                if (li < 0) { fprintf(stderr, "oracle: leader %d not in group\n", leader); abort(); }
                memmove(group + li, group + li + 1, sizeof(int) * (glen - li - 1)); glen--;      /* :328 */
                int nm = 0;
                members[nm++] = leader;
                if (vacancy > 1) {                                          /* :330 */
                    int k = vacancy - 1 < glen ? vacancy - 1 : glen;        /* :331 */
                    if (inj_nfol) k = inj_nfol[step];
                    for (int j = 0; j < k; j++) {
                        int pos;
                        if (inj_followers) {
                            int f = inj_followers[step * A + j];
                            pos = -1;
                            for (int i = 0; i < glen; i++) if (group[i] == f) pos = i;
                            if (pos < 0) { fprintf(stderr, "oracle: injected follower not in group\n"); abort(); }
                        } else {
                            pos = orc_below(orc_draw(seed_e, d, 2 + (uint64_t)j), glen);
                        }
                        members[nm++] = group[pos];
                        memmove(group + pos, group + pos + 1, sizeof(int) * (glen - pos - 1)); glen--; /* :332-333 */
                    }
                }
                for (int j = 0; j < nm; j++) orc_agent_step(e, members[j], action);             /* :338-340 */
                orc_task_update(e);                                         /* worker.py:74 */
                orc_agent_update(e);                                        /* :76 */
                if (rec_leader) rec_leader[step] = leader;
                if (rec_action) rec_action[step] = action;
                if (rec_nfol) rec_nfol[step] = nm - 1;
                if (rec_followers) { for (int j = 0; j < A; j++) rec_followers[step * A + j] = (int16_t)(j < nm - 1 ? members[j + 1] : -1); }
                if (rec_now) rec_now[step] = e->now;
                if (rec_mask) memcpy(rec_mask + step * (T + 1), mask, (size_t)T + 1);
                if (rec_agents) memcpy(rec_agents + step * 6 * A, ag, sizeof(float) * 6 * A);
                if (rec_tasks) memcpy(rec_tasks + step * 5 * (T + 1), tk, sizeof(float) * 5 * (T + 1));
                step++; d++;
            }
        }
        e->finished = orc_check_finished(e);                                /* :85 */
    }
    orc_finish_episode(e);                                                  /* :87 */
done:
    free(ids); free(gof); free(group); free(members); free(mask); free(ag); free(tk);
    return step;
}

/* ------------------------------------------------------------------ route replay */
/* env/task_env.py:595-599 */
void orc_pre_set_route(orc_env *e, int agent, const int32_t *actions, int n) {
    e->preset_none[agent] = 0;
    for (int i = 0; i < n; i++) iv_push(&e->preset[agent], actions[i]);
}

/* the four constants of env/task_env.py:567 / :221 (defaults 20, 20, 10, 100 = the reference) */
int orc_set_visibility(orc_env *e, int initial, int batch, int period, int cap) {
    if (initial < 0 || batch < 1 || period < 1 || cap < initial) return -1;
    e->vis_initial = initial; e->vis_batch = batch; e->vis_period = period; e->vis_cap = cap;
    return 0;
}

/* env/task_env.py:562-593 */
int orc_execute_by_route(orc_env *e, int reactive) {
    int32_t *ids = (int32_t *)malloc(sizeof(int32_t) * e->A);
    e->reactive = reactive;
    e->mwt = 100.0;                                                         /* :564 */
    int guard = 0;
    long steps = 0, step_cap = 64L * (e->A + e->T) + 4096; /* guard shared with the HIP kernel (not in the reference) */
    while (!e->finished && e->now < 200.0) {                                /* :565 */
        if (e->reactive) {                                                  /* :566-567 */
            double v = py_floordiv(e->now, (double)e->vis_period) * (double)e->vis_batch + (double)e->vis_initial;
            if (v < (double)e->vis_initial) v = (double)e->vis_initial;   /* np.clip(.., 20, 100) */
            if (v > (double)e->vis_cap) v = (double)e->vis_cap;
            e->visible_length = (int)v;
        }
        double t;
        int n = orc_next_decision(e, ids, &t);                              /* :568 */
        e->now = t;                                                         /* :569 */
        orc_task_update(e);                                                 /* :570 */
        orc_agent_update(e);                                                /* :571 */
        if (e->type_error) { free(ids); return -2; }
        if (n == 0) { if (++guard > 8) { e->truncated = 1; break; } } else guard = 0;
        for (int i = 0; i < n; i++) {                                       /* :572 */
            int a = ids[i];
            int remaining = e->preset[a].n - e->preset_head[a];
            int action;
            if (e->preset_none[a] || remaining == 0) action = 0;            /* :573-577 */
            else if (e->reactive && e->preset[a].v[e->preset_head[a]] > e->visible_length) action = 0; /* :578-584 */
            else action = e->preset[a].v[e->preset_head[a]++];              /* :585 pop(0) */
            orc_agent_step(e, a, action);
            if (++steps > step_cap) { e->truncated = 1; free(ids); return 0; }
            orc_task_update(e);                                             /* :575/:582/:586 */
            orc_agent_update(e);                                            /* :576/:583/:587 */
            if (e->type_error) { free(ids); return -2; }
        }
        e->finished = orc_check_finished(e);                                /* :588 */
    }
    free(ids);
    return 0;
}

/* ------------------------------------------------------------------ CPU baseline batch runner */
typedef struct {
    int B, A, T, episodes;
    const double *depot, *task_xy, *dur;
    const int32_t *req;
    const uint64_t *seeds;
    double *reward_out, *metrics_out;
    int64_t *steps_out;
    double *returns_out;     /* [B, episodes]: every episode's reward (nullable) */
    int32_t *n_finished_out; /* [B]: finished tasks of the last episode (nullable) */
    /* route replay (routes != NULL): execute_by_route instead of the RL loop */
    const int32_t *routes, *route_len;
    int route_cap, reactive;
    const int32_t *visibility; /* 4 ints or NULL */
    int32_t *status_out;       /* [B]: 0 ok, 1 truncated, 2 TypeError (nullable) */
    int next; /* work counter */
    int64_t total;
    pthread_mutex_t mu;
} batch_job;

static void *batch_worker(void *arg) {
    batch_job *j = (batch_job *)arg;
    orc_env *e = j->routes ? NULL : orc_create(j->A, j->T);
    int64_t local = 0;
    for (;;) {
        pthread_mutex_lock(&j->mu);
        int b = j->next < j->B ? j->next++ : -1;
        pthread_mutex_unlock(&j->mu);
        if (b < 0) break;
        if (j->routes) e = orc_create(j->A, j->T); /* the preset lists are consumed by the replay: a fresh env each */
        orc_load_instance(e, j->depot + 2 * (size_t)b, j->task_xy + 2 * (size_t)b * j->T, j->req + (size_t)b * j->T, j->dur + (size_t)b * j->T);
        int64_t steps = 0;
        int status = 0;
        if (j->routes) {
            if (j->visibility) orc_set_visibility(e, j->visibility[0], j->visibility[1], j->visibility[2], j->visibility[3]);
            for (int a = 0; a < j->A; a++) {
                int n = j->route_len[(size_t)b * j->A + a];
                if (n >= 0) orc_pre_set_route(e, a, j->routes + ((size_t)b * j->A + a) * j->route_cap, n);
            }
            int rc = orc_execute_by_route(e, j->reactive);
            if (rc == -2) status = 2;
            orc_finish_episode(e);
            for (int a = 0; a < j->A; a++) steps += e->route[a].n;       /* agent_step calls of the episode */
            if (e->truncated && !status) status = 1;
        } else {
            uint64_t d = 0;
            for (int ep = 0; ep < j->episodes; ep++) {
                if (ep) orc_clear_decisions(e);
                int64_t n = orc_rollout(e, j->seeds[b], d, ORC_POLICY_RANDOM, (int64_t)1 << 40, NULL, NULL, NULL, NULL,
                                        NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL);
                d += (uint64_t)n; steps += n;
                if (j->returns_out) { orc_summary s1; orc_summary_get(e, &s1); j->returns_out[(size_t)b * j->episodes + ep] = s1.reward; }
            }
            if (e->truncated) status = 1;
        }
        orc_summary s;
        orc_summary_get(e, &s);
        if (j->reward_out) j->reward_out[b] = s.reward;
        if (j->metrics_out) memcpy(j->metrics_out + 6 * (size_t)b, s.metrics, sizeof(double) * 6);
        if (j->steps_out) j->steps_out[b] = steps;
        if (j->n_finished_out) j->n_finished_out[b] = s.n_finished;
        if (j->status_out) j->status_out[b] = status;
        local += steps;
        if (j->routes) { orc_destroy(e); e = NULL; }
    }
    if (e) orc_destroy(e);
    pthread_mutex_lock(&j->mu);
    j->total += local;
    pthread_mutex_unlock(&j->mu);
    return NULL;
}

static int64_t batch_run(batch_job *j, int threads) {
    pthread_mutex_init(&j->mu, NULL);
    if (threads < 1) threads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * threads);
    for (int i = 0; i < threads; i++) pthread_create(&th[i], NULL, batch_worker, j);
    for (int i = 0; i < threads; i++) pthread_join(th[i], NULL);
    free(th);
    pthread_mutex_destroy(&j->mu);
    return j->total;
}

int64_t orc_batch_rollout_ex(int B, int A, int T, const double *depot, const double *task_xy, const int32_t *req,
                             const double *dur, const uint64_t *seeds, int episodes, int threads, double *reward_out,
                             int64_t *steps_out, double *metrics_out, double *returns_out, int32_t *n_finished_out) {
    batch_job j;
    memset(&j, 0, sizeof j);
    j.B = B; j.A = A; j.T = T; j.episodes = episodes; j.depot = depot; j.task_xy = task_xy; j.dur = dur; j.req = req; j.seeds = seeds;
    j.reward_out = reward_out; j.metrics_out = metrics_out; j.steps_out = steps_out; j.returns_out = returns_out;
    j.n_finished_out = n_finished_out;
    return batch_run(&j, threads);
}

int64_t orc_batch_rollout(int B, int A, int T, const double *depot, const double *task_xy, const int32_t *req,
                          const double *dur, const uint64_t *seeds, int episodes, int threads, double *reward_out,
                          int64_t *steps_out, double *metrics_out) {
    return orc_batch_rollout_ex(B, A, T, depot, task_xy, req, dur, seeds, episodes, threads, reward_out, steps_out, metrics_out, NULL, NULL);
}

int64_t orc_batch_replay(int B, int A, int T, const double *depot, const double *task_xy, const int32_t *req,
                         const double *dur, const int32_t *routes, const int32_t *route_len, int route_cap, int reactive,
                         const int32_t *visibility, int threads, double *reward_out, int64_t *steps_out, double *metrics_out,
                         int32_t *n_finished_out, int32_t *status_out) {
    batch_job j;
    memset(&j, 0, sizeof j);
    j.B = B; j.A = A; j.T = T; j.episodes = 1; j.depot = depot; j.task_xy = task_xy; j.dur = dur; j.req = req;
    j.routes = routes; j.route_len = route_len; j.route_cap = route_cap; j.reactive = reactive; j.visibility = visibility;
    j.reward_out = reward_out; j.metrics_out = metrics_out; j.steps_out = steps_out; j.n_finished_out = n_finished_out;
    j.status_out = status_out;
    return batch_run(&j, threads);
}
