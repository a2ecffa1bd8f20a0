/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not shipped, never on the product path.
 *
 * Rectangular linear-sum-assignment (minimisation), restating the published
 * algorithm behind scipy.optimize.linear_sum_assignment — the un-vendored
 * third-party dependency the reference calls at
 *   part_distillation/modeling/matcher.py:161   (scipy pinned ==1.8.1 in
 *   reference environment.yml; 1.15.3 in this image — same algorithm):
 * D. F. Crouse, "On implementing 2D rectangular assignment algorithms",
 * IEEE T-AES 52(4), 2016 — shortest augmenting paths with dual variables
 * (u, v), one augmentation per row; when there are more rows than columns the
 * transposed problem is solved.  Pinned against scipy itself in
 * tests/test_oracle_lsa.py (random, tied and degenerate cost matrices).
 *
 * cost: row-major double [nr, nc].  Outputs: row_ind/col_ind, min(nr,nc)
 * entries each, row_ind ascending (scipy's output convention).
 * Returns 0, or -1 if infeasible (an infinite/NaN cost blocks every path).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

static int solve_wide(int nr, int nc, const double *cost, int64_t *col4row)
{
  /* requires nr <= nc */
  double *u = calloc(nr, sizeof(double)), *v = calloc(nc, sizeof(double));
  double *dist = malloc(sizeof(double) * nc);
  int *pred = malloc(sizeof(int) * nc), *row4col = malloc(sizeof(int) * nc);
  int *todo = malloc(sizeof(int) * nc);
  char *row_seen = malloc(nr), *col_seen = malloc(nc);
  int status = 0;
  for (int j = 0; j < nc; ++j) row4col[j] = -1;
  for (int i = 0; i < nr; ++i) col4row[i] = -1;

  for (int cur = 0; cur < nr && status == 0; ++cur) {
    /* grow a shortest-path tree from row `cur` until an unassigned column is hit */
    double low = 0.0;
    int n_todo = nc, i = cur, sink = -1;
    for (int k = 0; k < nc; ++k) { todo[k] = nc - k - 1; dist[k] = INFINITY; col_seen[k] = 0; }
    for (int k = 0; k < nr; ++k) row_seen[k] = 0;
    while (sink < 0) {
      int pick = -1;
      double best = INFINITY;
      row_seen[i] = 1;
      for (int k = 0; k < n_todo; ++k) {
        int j = todo[k];
        double r = low + cost[(int64_t)i * nc + j] - u[i] - v[j];
        if (r < dist[j]) { dist[j] = r; pred[j] = i; }
        /* among equal distances prefer a column that ends the path */
        if (dist[j] < best || (dist[j] == best && row4col[j] < 0)) { best = dist[j]; pick = k; }
      }
      low = best;
      if (!(low < INFINITY) || pick < 0) { status = -1; break; }
      int j = todo[pick];
      if (row4col[j] < 0) sink = j; else i = row4col[j];
      col_seen[j] = 1;
      todo[pick] = todo[--n_todo];
    }
    if (status) break;
    /* dual update */
    u[cur] += low;
    for (int r = 0; r < nr; ++r)
      if (row_seen[r] && r != cur) u[r] += low - dist[col4row[r]];
    for (int j = 0; j < nc; ++j)
      if (col_seen[j]) v[j] -= low - dist[j];
    /* flip the assignments along the path */
    for (int j = sink;;) {
      int r = pred[j];
      row4col[j] = r;
      int64_t prev = col4row[r];
      col4row[r] = j;
      j = (int)prev;
      if (r == cur) break;
    }
  }
  free(u); free(v); free(dist); free(pred); free(row4col); free(todo); free(row_seen); free(col_seen);
  return status;
}

int pd_oracle_lsa(int nr, int nc, const double *cost, int64_t *row_ind, int64_t *col_ind)
{
  if (nr == 0 || nc == 0) return 0;
  for (int64_t k = 0; k < (int64_t)nr * nc; ++k)
    if (isnan(cost[k]) || cost[k] == -INFINITY) return -1;
  if (nr <= nc) {
    int st = solve_wide(nr, nc, cost, col_ind);
    for (int i = 0; i < nr; ++i) row_ind[i] = i;
    return st;
  }
  /* tall: solve the transpose, then report pairs ordered by original row */
  double *ct = malloc(sizeof(double) * (size_t)nr * nc);
  for (int i = 0; i < nr; ++i)
    for (int j = 0; j < nc; ++j) ct[(int64_t)j * nr + i] = cost[(int64_t)i * nc + j];
  int64_t *row4col = malloc(sizeof(int64_t) * nc); /* original row chosen for each original column */
  int st = solve_wide(nc, nr, ct, row4col);
  free(ct);
  if (st == 0) {
    /* selection-sort the nc pairs by row (nc is small on this path; stable for distinct rows) */
    char *used = calloc(nc, 1);
    for (int k = 0; k < nc; ++k) {
      int bj = -1;
      for (int j = 0; j < nc; ++j)
        if (!used[j] && (bj < 0 || row4col[j] < row4col[bj])) bj = j;
      used[bj] = 1;
      row_ind[k] = row4col[bj];
      col_ind[k] = bj;
    }
    free(used);
  }
  free(row4col);
  return st;
}
