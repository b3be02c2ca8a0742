/*
 * acn_qp.h -- C ABI of the MI355X batched MPC solver for adacharge.
 *
 * This is the drop-in boundary.  The reference has no FFI
 * at this path: its "solver call" is the Python statement
 *
 *     prob.solve(solver=self.solver, verbose=verbose)
 *         /root/reference/adacharge/adaptive_charging_optimization.py:318
 *
 * on a cvxpy Problem assembled at aco.py:220-284 and 315-317.  Each entry point
 * below names the piece of that path it replaces.  Plain C types only; every
 * buffer is owned by the caller; nothing is retained after a call returns
 * (for the *_device form: after the stream has drained).
 *
 * Problem solved, for every b in [0, batch):
 *
 *   minimise    1/2 pdiag_b |r|^2 + <q_b, r>                   r in R^{N x Tm}
 *   subject to  lb_b <= r <= ub_b                                (aco.py:61-79)
 *               sum_{t in [off, off+len)} r[i, t] <= cap          (aco.py:105-123;
 *                   (== cap when s_eq_b)                            cap in A-periods)
 *               for every period t, rows of the site matrix G:
 *                 LINEAR  (G r[:, t])_j <= limits_j               (aco.py:165-172)
 *                 SOC     |((G r)_j, (G r)_{j+M})|_2 <= limits_j  (aco.py:151-164)
 *                 peak    sum_i r[i, t] <= peak_b[t]              (aco.py:196-198)
 *   plus, when the site carries a "flat" row v = voltages / 1e3 (kW per A), the objective term
 *               1/2 lf_b sum_t (v' r[:, t])^2                     (load_flattening, aco.py:403-408;
 *                                                                  its linear part is already in q)
 *   and, when it carries a "max" row (same v), the objective term
 *               dc_b * max(max_t v' r[:, t], dfloor_b)            (demand_charge / peak, aco.py:387-400)
 *
 * Layouts are C order: r, lb, ub, q are [batch][N][Tm] -- the (N, T) rates
 * matrix the reference returns at aco.py:321, one per problem.
 */
#ifndef ACN_QP_H
#define ACN_QP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ACNQP_ABI_VERSION 9

/* cone of the infrastructure rows (constraint_type at aco.py:35, 151, 165) */
#define ACNQP_CONE_LINEAR 0
#define ACNQP_CONE_SOC 1

/* per-problem status; the Python layer maps anything but SOLVED to
 * InfeasibilityException exactly as aco.py:319-320 does for cvxpy statuses */
#define ACNQP_STATUS_UNSET 0
#define ACNQP_STATUS_SOLVED 1             /* cp.OPTIMAL                     */
#define ACNQP_STATUS_MAX_ITER 2           /* residuals above tolerance at max_iter, or stalled above the
                                             SOLVED_INACCURATE level (no 10 % progress for options.stall_iters
                                             iterations), in every pass (options.retry_passes)              */
#define ACNQP_STATUS_PRIMAL_INFEASIBLE 3  /* ADMM certificate (cp.INFEASIBLE) */
#define ACNQP_STATUS_EMPTY_SET 4          /* a session's bounds cannot meet its energy row */
#define ACNQP_STATUS_SOLVED_INACCURATE 5   /* max_iter or the stall rule ended every pass short of the tolerance,
                                             the best one with both residuals within 100x their tolerance or
                                             within options.inaccurate_floor, whichever is looser:
                                             cp.OPTIMAL_INACCURATE, which the reference accepts (aco.py:319) */

/* return codes (never C++ exceptions across the ABI) */
#define ACNQP_OK 0
#define ACNQP_ERR_INVALID (-1)     /* bad argument / unsupported shape */
#define ACNQP_ERR_HIP (-2)         /* HIP runtime failure              */
#define ACNQP_ERR_NO_DEVICE (-3)

typedef struct acnqp_handle acnqp_handle; /* opaque; one per (site, GPU) */

/* Site data shared by every problem of a batch: what InfrastructureInfo
 * contributes to aco.py:126-198.  G is [n_rows][n_evse] row-major:
 *   LINEAR: |constraint_matrix| (n_infra rows)            aco.py:171
 *   SOC:    [C*cos(phi); C*sin(phi)] (2*n_infra rows)     aco.py:156-158
 *   then one row voltages/1e3 iff has_flat               aco.py:336-344, 406
 *   then one row voltages/1e3 iff has_max                aco.py:387-400
 *   then one all-ones row iff has_peak                   aco.py:197          */
typedef struct {
  int32_t n_evse;        /* N                                   */
  int32_t n_infra;       /* M: rows of constraint_matrix         */
  int32_t n_rows;        /* rows of G = M or 2M, + has_flat + has_max + has_peak */
  int32_t cone;          /* ACNQP_CONE_*                         */
  int32_t has_peak;      /* 0 / 1                                */
  int32_t has_flat;      /* 0 / 1: aggregate-power row for load_flattening */
  int32_t has_max;       /* 0 / 1: aggregate-power row for demand_charge / peak */
  const double* G;       /* [n_rows * n_evse]                    */
  const double* limits;  /* [n_infra]  constraint_limits         */
} acnqp_site;

/* One batch of structured problems.  Pointers are HOST pointers for
 * acnqp_solve_batch and DEVICE pointers for acnqp_solve_batch_device. */
typedef struct {
  int32_t batch;           /* B                                              */
  int32_t t_max;           /* Tm: padded horizon of every array below        */
  int32_t k_sessions;      /* K: session slots per EVSE (>= 1; disjoint windows
                              per EVSE; offline instances, adacharge.py:249-276,
                              reach tens of sessions per EVSE)                */
  const int32_t* horizon;  /* [B]        own horizon T_b <= Tm (aco.py:243)  */
  const double* lb;        /* [B*N*Tm]   0 outside session windows           */
  const double* ub;        /* [B*N*Tm]   finite                              */
  const double* q;         /* [B*N*Tm]   linear cost (minimisation form)     */
  const double* pdiag;     /* [B]        P = pdiag * I  (2 * equal_share)    */
  const int32_t* s_off;    /* [B*K*N]    window start per (slot, EVSE)       */
  const int32_t* s_len;    /* [B*K*N]    window length, 0 = empty slot       */
  const double* s_cap;     /* [B*K*N]    energy cap in A-periods             */
  const uint8_t* s_eq;     /* [B]        1: energy rows are equalities       */
  const double* peak;      /* [B*Tm] or NULL; +inf = unlimited period        */
  const double* lf;        /* [B] or NULL: weight of 1/2 lf (v' r_t)^2 (2 * load_flattening coefficient) */
  const double* dc;        /* [B] or NULL: weight (>= 0) of max(max_t v' r_t, dfloor)  [$ per kW]     */
  const double* dfloor;    /* [B] or NULL: previous / baseline peak in kW (aco.py:390-394)           */
  /* Optional warm start (both or neither; NULL = cold, which is what the reference does: nothing survives a
   * schedule() call, adacharge.py:152-158).  A closed-loop caller passes the previous step's schedule and
   * site-row multipliers (results.y), both shifted by the periods that have elapsed.  The solve then starts from
   * z = Proj(warm_x), y2 = warm_y and the multipliers of the box / energy set that make the pair stationary,
   * y1 = -(P z + q + G' y2).  The optimum does not depend on it; the iteration count does.                       */
  const double* warm_x;    /* [B*N*Tm] or NULL                                                        */
  const double* warm_y;    /* [B*n_rows*Tm] or NULL: multipliers of the rows of acnqp_site.G, per period */
} acnqp_problems;

typedef struct {
  double* x;         /* [B*N*Tm]  schedule (the feasible ADMM iterate z)      */
  int32_t* status;   /* [B]       ACNQP_STATUS_*                              */
  int32_t* iters;    /* [B]                                                   */
  double* pri_res;   /* [B]       |A r - z|_inf at exit                       */
  double* dua_res;   /* [B]       |P r + q + A'y|_inf at exit                 */
  double* obj;       /* [B]       1/2 pdiag |x|^2 + <q, x>                    */
  double* y;         /* [B*n_rows*Tm] or NULL (not wanted): multipliers of the rows of acnqp_site.G at exit
                        (row order and units of G; a SOC pair's two rows carry the pair's two components)
                        -- the warm_y of a later, similar problem                                 */
  double* x_dev;     /* host-buffer entry points only, optional (NULL = not wanted): a DEVICE
                        pointer [B*N*Tm] on the handle's GPU that also receives the schedules,
                        so that a collective (the RCCL all-gather of a multi-GPU job) can start
                        from HBM without re-uploading them.  Ignored by acnqp_solve_batch_device. */
} acnqp_results;

/* Solver options -- the knobs cvxpy would forward to its solver (the
 * reference sets none, aco.py:318; defaults: acnqp_default_options). */
typedef struct {
  double eps_abs;        /* absolute residual tolerance                       */
  double eps_rel;        /* relative residual tolerance                       */
  int32_t max_iter;
  int32_t check_every;   /* residual check period (iterations)                */
  int32_t adapt_every;   /* rho adaptation period, 0 = fixed rho              */
  double rho;            /* initial ADMM penalty                              */
  double sigma;          /* proximal weight on x                              */
  double alpha;          /* over-relaxation in (0, 2)                         */
  double adapt_tol;      /* adapt when the residual ratio leaves [1/tol, tol];
                            the band widens by tol/8 with every adaptation made,
                            so the penalty cannot cycle                          */
  double reg_rel;        /* scale-free Tikhonov floor: effective pdiag =
                            max(pdiag, reg_rel * |q|_inf / (max(ub) * T_b)); 0 disables.
                            On LP instances a small floor returns the least-norm
                            LP optimum (exact regularisation)  */
  int32_t precision;     /* 64 or 32: arithmetic type of the ADMM loop        */
  int32_t accel_mem;     /* Anderson acceleration of the ADMM fixed-point map: columns of
                            history requested (0 = plain ADMM).  The kernels use
                            min(accel_mem, what fits their LDS for the problem shape);
                            acnqp_accel_columns reports that number                */
  /* -- ABI v7 ------------------------------------------------------------------------------------------ */
  int32_t stall_iters;   /* stall rule: a pass whose residual score max(pri / eps_pri, dua / eps_dua) has not
                            improved by 10 % for this many iterations, and sits within 1.25x of its best, ends
                            (SOLVED_INACCURATE if it qualifies, MAX_ITER otherwise) instead of burning max_iter
                            iterations on a plateau.  0 = off.  Default 3000 (the longest wait between two
                            improvements seen on any converging instance of tools/ and tests/ is 1,240)     */
  int32_t retry_passes;  /* a problem whose pass ends MAX_ITER / SOLVED_INACCURATE after >= stall_iters (3000 if
                            the stall rule is off) iterations is solved again from a COLD start with a FIXED
                            penalty retry_rho * 4^(pass - 1) -- inside the same kernel launch, for every entry
                            point -- up to this many times; the best pass is returned (SOLVED > SOLVED_INACCURATE
                            > MAX_ITER, the first of equals), iters is the total.  0 = single pass.  Default 2.
                            Only when adapt_every > 0 (a caller who fixed the penalty gets that penalty only).
                            Why it works: DESIGN.md section 2 (the plateau is the adaptive penalty's doing)  */
  int32_t retry_max_iter;/* iteration limit of a retry pass (min with max_iter).  Default 8000            */
  /* -- ABI v8 ------------------------------------------------------------------------------------------ */
  int32_t polish_iters;  /* a problem of a small site (N <= 64, horizon <= 32, <= 4 sessions per EVSE, no load-flattening /
                            demand-charge row, cold start) that has not converged after this many ADMM iterations is
                            handed to the POLISH: an active-set Newton method on the KKT conditions that starts from the
                            working set the ADMM iterate suggests (variables on a bound, tight energy rows, site rows with a
                            non-zero multiplier) -- a few dozen small dense solves in one workgroup, inside the same call.
                            Its answer is accepted only with the KKT conditions verified on the full problem (status
                            SOLVED, residuals of that check in pri_res / dua_res, iters = ADMM iterations + Newton
                            rounds); otherwise the problem is solved as if there were no polish (from scratch, with the
                            retry passes).  What an interior-point method -- the reference's ECOS, aco.py:318 -- gives
                            for free: no plateau on the congested, tangentially degenerate instances.  0 = off.
                            Default 800                                                                      */
  double retry_rho;      /* penalty of the first retry pass.  Default 0.5                                 */
  double inaccurate_floor; /* residual tolerance (absolute and relative) below which a pass that ran out of
                            iterations still counts as SOLVED_INACCURATE even when 100x the requested tolerance
                            is tighter.  Default 1e-5 (what cvxpy hands OSQP as eps_abs = eps_rel).  0 = the
                            100x rule alone                                                               */
  /* -- ABI v9 ------------------------------------------------------------------------------------------ */
  int32_t polish_stall;  /* early hand-over to the polish (one-wave-per-problem kernel: N <= 64, horizon <= 12 with up to 16
                            site rows): from polish_iters / 2 on, a problem whose residual score has not improved by 10 % for
                            this many iterations goes to the polish at once instead of at polish_iters.  For a launch in
                            which every problem has a wavefront of its own -- up to 1,024 per GPU: the scenario MPC of
                            BASELINE configs[3] -- the launch lasts as long as its slowest problem, and 100 takes a
                            quarter off it (1,024 scenarios of one site: 3.4 -> 2.5 ms, 9.8 -> 7.3 ms).  On a
                            throughput-bound launch it sends ten times as many problems to the polish, most of which
                            the ADMM would have finished by itself: slower (one 256-batch per call 81 -> 62 k QP/s).
                            Same optimum either way (the polish verifies the KKT conditions).  0 = off.  Default 0 */
} acnqp_options;

/* acnqp_create -- uploads the site once.  Replaces the per-call rebuilding of
 * the infrastructure atoms at aco.py:157-172 (the reference re-creates them on
 * every MPC step, adacharge.py:152-158).  device_id: HIP ordinal. */
int acnqp_create(const acnqp_site* site, int32_t device_id, acnqp_handle** out);

/* acnqp_solve_batch -- host buffers in, host buffers out, synchronous.
 * Replaces cp.Problem(...).solve(...) at aco.py:315-318 for B problems.
 * Same as acnqp_solve_batches with one batch: a large batch is cut into chunks whose
 * H2D copies, kernels and D2H copies overlap.                                       */
int acnqp_solve_batch(acnqp_handle* h, const acnqp_problems* p,
                      const acnqp_options* o, acnqp_results* r);

/* acnqp_solve_batches -- n_batches independent host-buffer batches (p[g] -> r[g]) in
 * ONE pipelined pass, synchronous.  What a caller with many batches of MPC snapshots
 * (timesteps, sites, demand scenarios: the reference would loop over aco.py:286-321)
 * submits at once: consecutive batches of one shape (t_max, k_sessions) share kernel
 * launches of a few thousand problems, so that a launch does not idle on its slowest
 * problem; chunks rotate over internal streams with their own device staging, so that
 * the H2D copies of chunk c+1 and the D2H copies of chunk c-1 overlap chunk c's kernel.
 * Buffers allocated with acnqp_host_alloc (pinned) are copied by DMA without a staging
 * hop; pageable buffers work, slower.  Every pointer is a HOST pointer; nothing is
 * retained after return.  Every problem's status is set (never ACNQP_STATUS_UNSET) or
 * the call fails with ACNQP_ERR_HIP.                                                 */
int acnqp_solve_batches(acnqp_handle* h, int32_t n_batches, const acnqp_problems* p,
                        const acnqp_options* o, acnqp_results* r);

/* A batch of problems as the SESSION TABLE the reference's statement is made of, instead of the dense arrays of
 * acnqp_problems: what charging_rate_bounds (aco.py:45-79) and energy_constraints (aco.py:81-124) loop over -- one
 * record per active session -- plus the linear cost once per distinct horizon (build_objective, aco.py:200-218: it
 * depends on a problem only through T = max(offset + remaining), aco.py:243-245).  The library forms lb, ub, q and
 * the per-EVSE session slots ON THE DEVICE (acn_qp_api.hip: table_expand_*), so that a problem costs ~1-4 KB of
 * host-to-device traffic instead of 21.6 KB at 54 x 12 and no caller ever fills an (N, T) array per problem.
 * Sessions are grouped by problem: problem b owns the sessions [sess_seg[b], sess_seg[b + 1]); session s owns the
 * rate entries [rate_seg[s], rate_seg[s + 1]) -- exactly s_len[s] of them (aco.py:68, 73).  Windows of one EVSE must
 * be disjoint (s_slot numbers them 0 .. k_sessions - 1 per EVSE and problem).  All pointers are HOST pointers.      */
typedef struct {
  int32_t batch;            /* B                                                                       */
  int32_t t_max;            /* Tm >= every horizon                                                     */
  int32_t k_sessions;       /* K: session slots per EVSE                                               */
  int32_t n_sessions;       /* S                                                                       */
  int32_t n_horizons;       /* H: rows of q_table                                                      */
  const int32_t* horizon;   /* [B]       T_b                                                           */
  const int32_t* q_index;   /* [B]       row of q_table holding this problem's linear cost             */
  const double* q_table;    /* [H*N*Tm]  linear cost (minimisation form) per distinct horizon          */
  const double* pdiag;      /* [B]                                                                     */
  const uint8_t* s_eq;      /* [B]                                                                     */
  const double* peak;       /* [B*Tm] or NULL                                                          */
  const double* lf;         /* [B] or NULL                                                             */
  const double* dc;         /* [B] or NULL                                                             */
  const double* dfloor;     /* [B] or NULL                                                             */
  const int32_t* sess_seg;  /* [B+1]     sessions of problem b: [sess_seg[b], sess_seg[b+1])           */
  const int32_t* s_evse;    /* [S]       EVSE index                                                    */
  const int32_t* s_slot;    /* [S]       slot of the session among its EVSE's sessions (0 .. K-1)      */
  const int32_t* s_off;     /* [S]       arrival_offset                                                */
  const int32_t* s_len;     /* [S]       remaining_time (<= 0: no window, no energy row)               */
  const double* s_cap;      /* [S]       remaining_demand in A-periods (aco.py:114)                    */
  const int32_t* rate_seg;  /* [S+1]     rate entries of session s: [rate_seg[s], rate_seg[s+1])       */
  const double* min_rates;  /* [rate_seg[S]]  aco.py:68                                                */
  const double* max_rates;  /* [rate_seg[S]]  aco.py:73 (ub < lb -> lb, aco.py:75, is applied here)     */
} acnqp_table;

/* acnqp_solve_table -- acnqp_solve_batch for a session table: same pipeline (chunks over internal streams, copies
 * overlapped with kernels), same kernels, same results bit for bit as the dense arrays the table stands for
 * (tests/test_table_entry.py).  Cold start only; results as acnqp_solve_batch (r->y and r->x_dev honoured).        */
int acnqp_solve_table(acnqp_handle* h, const acnqp_table* t, const acnqp_options* o, acnqp_results* r);

/* Pinned (page-locked) host memory for problem / result arrays; NULL on failure.    */
void* acnqp_host_alloc(size_t bytes);
void acnqp_host_free(void* p);

/* acnqp_solve_batch_device -- same, but every pointer in *p and *r is a device
 * pointer on the handle's GPU and the work is enqueued on `hip_stream`
 * (a hipStream_t, NULL = default stream); returns without synchronising.
 * r->status is cleared to ACNQP_STATUS_UNSET on the stream before the launch: a
 * problem still UNSET after synchronisation was never processed.              */
int acnqp_solve_batch_device(acnqp_handle* h, const acnqp_problems* p,
                             const acnqp_options* o, acnqp_results* r,
                             void* hip_stream);

/* acnqp_destroy -- frees device copies of the site and the staging workspace. */
void acnqp_destroy(acnqp_handle* h);

void acnqp_default_options(acnqp_options* o);

/* Text of the last error on the calling thread ("" if none). */
const char* acnqp_last_error(void);

int32_t acnqp_abi_version(void);

/* Duration in milliseconds of the most recent kernel launched through this
 * handle, measured with HIP events on the launch stream (valid after the
 * stream has been synchronised); < 0 if none.  Used by bench.py's roofline. */
float acnqp_last_kernel_ms(acnqp_handle* h);

/* Durations (ms, oldest first) of the launches made through this handle since
 * the previous call -- at most the 64 most recent and at most `capacity` --
 * without forcing the caller to synchronise between launches (batches kept in
 * flight on several streams).  Blocks until those launches have finished.
 * Returns the number of values written.                                      */
int32_t acnqp_kernel_times(acnqp_handle* h, float* out_ms, int32_t capacity);

/* Kernel launches made through this handle since it was created (acnqp_kernel_times keeps the 64 most recent
 * durations: a caller that sums them can tell from this count whether any were dropped).                    */
int64_t acnqp_launch_count(acnqp_handle* h);

/* Of those, the launches whose QUEUE ORDER was sorted (longest expected problem first, by session count: launches of
 * >= 768 problems on sites without a load-flattening / demand-charge row).  Test plumbing: lets a test assert that the
 * ordered path really ran.  Every launch hands its problems to the resident workgroups through a work queue (one
 * atomic counter per launch); ACNQP_NO_QUEUE=1 / ACNQP_NO_ORDER=1 in the environment select the static schedule / the
 * natural order (diagnostics: results do not depend on either).                                                    */
int64_t acnqp_ordered_launch_count(acnqp_handle* h);

/* Counters of the polish over the handle's life (synchronises the device): out[0] problems handed to the polish,
 * out[1] solved by it, out[2..5] given up because of: more tight site rows than its LDS holds, a non-positive pivot,
 * the round limit, the final KKT check; out[6] Newton rounds made; out[8..15] time per phase of the polish kernel in
 * 10 ns ticks, summed over its workgroups (rows + gradient, Schur matrix, Cholesky, triangular solves, step, ratio test,
 * update, multiplier check + verification).  Returns ACNQP_OK.  Bench / test plumbing.                               */
int acnqp_polish_stats(acnqp_handle* h, int64_t* out, int32_t capacity);

/* Anderson columns the kernels will actually use for problems of this shape
 * (t_max periods, k_sessions slots) at the given precision when `requested`
 * columns are asked for: a function of the shape only, never of the batch
 * size (the long-horizon, large-site and general-shape kernels keep their
 * ring in global memory: 5).  No reference equivalent:
 * test/bench plumbing so that a CPU restatement can run the same algorithm.  */
int32_t acnqp_accel_columns(acnqp_handle* h, int32_t t_max, int32_t k_sessions, int32_t precision,
                            int32_t requested);

#ifdef __cplusplus
}
#endif
#endif /* ACN_QP_H */
