/* timing_oracle.c — plain-C restatement of whisper/timing.py:82-103 (dtw_cpu fill) and :19-54
 * (median_filter, sort path).  Test infrastructure only (see oracle/__init__.py). */
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* x: float32 [N][M]; trace: int8 [N+1][M+1] (interior filled; caller sets the borders) */
void oracle_dtw_trace(const float *x, int N, int M, signed char *trace) {
  float *cost = (float *)malloc((size_t)(N + 1) * (M + 1) * sizeof(float));
  for (long i = 0; i < (long)(N + 1) * (M + 1); ++i) cost[i] = INFINITY;
  cost[0] = 0.0f;
  const long W = M + 1;
  for (int j = 1; j <= M; ++j) {              /* loop order of timing.py:89-90 */
    for (int i = 1; i <= N; ++i) {
      const float c0 = cost[(i - 1) * W + j - 1], c1 = cost[(i - 1) * W + j], c2 = cost[i * W + j - 1];
      float c; signed char t;
      if (c0 < c1 && c0 < c2) { c = c0; t = 0; }
      else if (c1 < c0 && c1 < c2) { c = c1; t = 1; }
      else { c = c2; t = 2; }
      /* x is float64 in the reference (x.double(), timing.py:151), cost is float32 (:84) */
      cost[i * W + j] = (float)((double)x[(long)(i - 1) * M + j - 1] + (double)c);
      trace[i * W + j] = t;
    }
  }
  free(cost);
}

static int cmp_float(const void *a, const void *b) {
  const float x = *(const float *)a, y = *(const float *)b;
  return (x > y) - (x < y);
}

void oracle_median_filter(const float *x, float *out, long rows, int n, int width) {
  const int pad = width / 2;
  float *win = (float *)malloc(sizeof(float) * width);
  for (long r = 0; r < rows; ++r) {
    const float *xr = x + r * n;
    for (int i = 0; i < n; ++i) {
      for (int j = 0; j < width; ++j) {
        int k = i - pad + j;
        if (k < 0) k = -k;                     /* F.pad(mode="reflect"), timing.py:35 */
        if (k >= n) k = 2 * (n - 1) - k;
        win[j] = xr[k];
      }
      qsort(win, width, sizeof(float), cmp_float);
      out[r * n + i] = win[pad];
    }
  }
  free(win);
}
