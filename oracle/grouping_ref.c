/* Oracle (TEST INFRASTRUCTURE, see oracle/__init__.py): plain-C restatement of the
 * tf_ops/grouping kernels of the reference.
 *
 *   ref_query_ball_point   tf_ops/grouping/tf_grouping_g.cu:3-36   (CPU twin test/query_ball_point.cpp:19-47)
 *   ref_group_point        tf_ops/grouping/tf_grouping_g.cu:40-57  (test/query_ball_point.cpp:52-66)
 *   ref_group_point_grad   tf_ops/grouping/tf_grouping_g.cu:61-78  (test/query_ball_point.cpp:70-84)
 *   ref_selection_sort     tf_ops/grouping/tf_grouping_g.cu:83-123 (test/selection_sort.cpp:20-63)
 *   ref_knn_dist           tf_ops/grouping/tf_grouping.py:48-73    (dist = reduce_sum((xyz1-xyz2)^2,-1))
 *   ref_nn1_fill           test_region_grow.py:308-316             (NumPy float32 pairwise row sum)
 *
 * Validated against the reference's own C++ functions compiled into oracle/_ref/ (Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

void ref_query_ball_point(int b, int n, int m, float radius, int nsample,
                          const float *xyz1, const float *xyz2, int *idx, int *pts_cnt) {
    for (int i = 0; i < b; ++i) {
        for (int j = 0; j < m; ++j) {
            int cnt = 0;
            for (int k = 0; k < n && cnt < nsample; ++k) {
                float dx = xyz2[j * 3 + 0] - xyz1[k * 3 + 0];
                float dy = xyz2[j * 3 + 1] - xyz1[k * 3 + 1];
                float dz = xyz2[j * 3 + 2] - xyz1[k * 3 + 2];
                float d = fmaxf(sqrtf(dx * dx + dy * dy + dz * dz), 1e-20f);
                if (d < radius) {
                    if (cnt == 0)
                        for (int l = 0; l < nsample; ++l) idx[j * nsample + l] = k;
                    idx[j * nsample + cnt] = k;
                    cnt += 1;
                }
            }
            pts_cnt[j] = cnt;
        }
        xyz1 += n * 3; xyz2 += m * 3; idx += m * nsample; pts_cnt += m;
    }
}

void ref_group_point(int b, int n, int c, int m, int nsample, const float *points, const int *idx, float *out) {
    for (int i = 0; i < b; ++i) {
        for (int j = 0; j < m; ++j)
            for (int k = 0; k < nsample; ++k) {
                int ii = idx[j * nsample + k];
                for (int l = 0; l < c; ++l) out[(j * nsample + k) * c + l] = points[ii * c + l];
            }
        points += n * c; idx += m * nsample; out += m * nsample * c;
    }
}

void ref_group_point_grad(int b, int n, int c, int m, int nsample, const float *grad_out, const int *idx,
                          float *grad_points) {
    for (int i = 0; i < b; ++i) {
        for (int j = 0; j < m; ++j)
            for (int k = 0; k < nsample; ++k) {
                int ii = idx[j * nsample + k];
                for (int l = 0; l < c; ++l) grad_points[ii * c + l] += grad_out[(j * nsample + k) * c + l];
            }
        idx += m * nsample; grad_out += m * nsample * c; grad_points += n * c;
    }
}

void ref_selection_sort(int b, int n, int m, int k, const float *dist, int *outi, float *out) {
    for (long long r = 0; r < (long long)b * m; ++r) {
        const float *d = dist + r * n;
        float *o = out + r * n;
        int *oi = outi + r * n;
        for (int s = 0; s < n; ++s) { o[s] = d[s]; oi[s] = s; }
        for (int s = 0; s < k; ++s) {
            int mn = s;
            for (int t = s + 1; t < n; ++t)
                if (o[t] < o[mn]) mn = t;
            if (mn != s) {
                float tf = o[mn]; o[mn] = o[s]; o[s] = tf;
                int ti = oi[mn]; oi[mn] = oi[s]; oi[s] = ti;
            }
        }
    }
}

/* dist[b,m,n] = sum_c (xyz1[b,n,c] - xyz2[b,m,c])^2, float32, summed in channel order. */
void ref_knn_dist(int b, int n, int m, int c, const float *xyz1, const float *xyz2, float *dist) {
    for (int i = 0; i < b; ++i)
        for (int j = 0; j < m; ++j)
            for (int k = 0; k < n; ++k) {
                float s = 0.f;
                for (int l = 0; l < c; ++l) {
                    float d = xyz1[((long long)i * n + k) * c + l] - xyz2[((long long)i * m + j) * c + l];
                    float sq = d * d;
                    s = (l == 0) ? sq : s + sq;
                }
                dist[((long long)i * m + j) * n + k] = s;
            }
}

/* NumPy's float32 add.reduce over a contiguous row of F elements (pairwise_sum, F < 128):
 * F < 8: left-to-right; else 8 partial sums over i%8 for the leading 8*(F/8) entries,
 * combined ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the tail added left-to-right. */
static float np_rowsum_f32(const float *a, int F) {
    if (F < 8) {
        float s = a[0];   /* numpy seeds the reduction with the first element */
        for (int i = 1; i < F; ++i) s += a[i];
        return s;
    }
    float r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int i;
    for (i = 8; i < F - (F % 8); i += 8)
        for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    float s = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < F; ++i) s += a[i];
    return s;
}

/* label[i]==0 -> label of the labeled point with the smallest sum_F (p_j - p_i)^2 (first min). */
void ref_nn1_fill(int N, int F, const float *points, const int *label_in, int *label_out) {
    float tmp[64];
    for (int i = 0; i < N; ++i) {
        label_out[i] = label_in[i];
        if (label_in[i] != 0) continue;
        float best = INFINITY; int bj = -1;
        for (int j = 0; j < N; ++j) {
            if (label_in[j] == 0) continue;
            for (int l = 0; l < F; ++l) {
                volatile float d = points[(long long)j * F + l] - points[(long long)i * F + l];
                volatile float sq = d * d;
                tmp[l] = sq;
            }
            float s = np_rowsum_f32(tmp, F);
            if (bj < 0 || s < best) { best = s; bj = j; }
        }
        if (bj >= 0) label_out[i] = label_in[bj];
    }
}
