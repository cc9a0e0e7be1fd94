// Oracle (TEST INFRASTRUCTURE): extern "C" doors onto the REFERENCE's own CPU functions.
// The reference sources are compiled from where they lie under /root/reference (paths passed by
// oracle/Makefile as -DREF_SRC=...); nothing is copied.  Outputs go to oracle/_ref/ only.
//   WHICH=1: tf_ops/grouping/test/query_ball_point.cpp  (query_ball_point_cpu :19-47,
//            group_point_cpu :52-66, group_point_grad_cpu :70-84)
//   WHICH=2: tf_ops/grouping/test/selection_sort.cpp    (selection_sort_cpu :20-63; prints to stdout)
#define main ref_unused_main
#include REF_SRC
#undef main

#if WHICH == 1
extern "C" void refcpu_query_ball_point(int b, int n, int m, float radius, int nsample, const float *xyz1,
                                        const float *xyz2, int *idx) {
    query_ball_point_cpu(b, n, m, radius, nsample, xyz1, xyz2, idx);
}
extern "C" void refcpu_group_point(int b, int n, int c, int m, int nsample, const float *points, const int *idx,
                                   float *out) {
    group_point_cpu(b, n, c, m, nsample, points, idx, out);
}
extern "C" void refcpu_group_point_grad(int b, int n, int c, int m, int nsample, const float *grad_out,
                                        const int *idx, float *grad_points) {
    group_point_grad_cpu(b, n, c, m, nsample, grad_out, idx, grad_points);
}
#else
extern "C" void refcpu_selection_sort(int b, int n, int m, int k, const float *dist, int *idx, float *val) {
    selection_sort_cpu(b, n, m, k, dist, idx, val);
}
extern "C" int refcpu_selection_sort_main() { return ref_unused_main(); }
#endif
