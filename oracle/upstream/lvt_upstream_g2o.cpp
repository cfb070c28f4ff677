// lvt_upstream_g2o.cpp -- OPTIONAL pin of the oracle's motion-only BA against the real g2o.  TEST INFRASTRUCTURE ONLY.
//
// The reference refines the pose with g2o as configured at lvt_pnp_solver.cpp:44-53 (SparseOptimizer + Levenberg over BlockSolver_6_3 with
// the PCG linear solver) and :60-128 (one free VertexCam, one fixed VertexSBAPointXYZ + EdgeProjectP2MC with a Cauchy kernel of width
// sqrt(5.991) per match, two passes of optimize(5), edges with chi2 > 5.991 moved to level 1 after each pass).  g2o is not vendored under
// /root/reference and not installed here or on the GPU box, so the oracle restates that schedule from SURVEY A.6 and stays unpinned.  This
// file calls the same g2o entry points with the same arguments behind a C ABI; `make -C oracle upstream` builds it where
// <g2o/core/sparse_optimizer.h> exists, and tests/test_upstream_pin.py::test_g2o_motion_only_ba then holds the oracle's pose, inlier marks
// and per-edge chi2 to it.  Written for the API of the pinned tag (20170730: solvers handed over as raw pointers); define
// LVT_G2O_UNIQUE_PTR for releases that take std::unique_ptr.
#if !defined(__has_include)
#error "this adapter needs a compiler with __has_include"
#endif
#if !__has_include(<g2o/core/sparse_optimizer.h>) || !__has_include(<g2o/types/sba/types_sba.h>)
#error "g2o is not installed: the upstream pin of the pose refinement cannot be built here (expected in this image)"
#endif

#include <g2o/core/block_solver.h>
#include <g2o/core/optimization_algorithm_levenberg.h>
#include <g2o/core/robust_kernel_impl.h>
#include <g2o/core/sparse_optimizer.h>
#include <g2o/solvers/pcg/linear_solver_pcg.h>
#include <g2o/types/sba/types_sba.h>

#include <cmath>
#include <memory>
#include <vector>

namespace {
using PoseBlock = g2o::BlockSolver_6_3::PoseMatrixType;

g2o::OptimizationAlgorithmLevenberg *make_algorithm() {
#ifdef LVT_G2O_UNIQUE_PTR
    auto lin = std::make_unique<g2o::LinearSolverPCG<PoseBlock>>();
    return new g2o::OptimizationAlgorithmLevenberg(std::make_unique<g2o::BlockSolver_6_3>(std::move(lin)));
#else
    return new g2o::OptimizationAlgorithmLevenberg(new g2o::BlockSolver_6_3(new g2o::LinearSolverPCG<PoseBlock>()));
#endif
}
}  // namespace

extern "C" {

// pose in / out: quaternion (w, x, y, z) + position, camera-to-world; pts n x 3 f64 (world), obs n x 2 f32 (pixels).
// inlier_marks[n]: 1 = never demoted; chi2_out[n]: the edge's chi2() after the second pass's gate; returns the number of inliers.
__attribute__((visibility("default"))) int lvtu_g2o_pnp(double fx, double fy, double cx, double cy, double baseline, const double q_in[4],
                                                        const double p_in[3], const double *pts, const float *obs, int n, double q_out[4],
                                                        double p_out[3], int *inlier_marks, double *chi2_out) {
    const double th2 = 5.991;                 // lvt_definitions.h: LVT_REPROJECTION_TH2
    const double kernel_width = std::sqrt(th2);
    g2o::SparseOptimizer graph;
    graph.setVerbose(false);
    graph.setAlgorithm(make_algorithm());

    g2o::SBACam cam0(Eigen::Quaterniond(q_in[0], q_in[1], q_in[2], q_in[3]), Eigen::Vector3d(p_in[0], p_in[1], p_in[2]));
    cam0.setKcam(fx, fy, cx, cy, baseline);
    auto *cam = new g2o::VertexCam();
    cam->setId(0);
    cam->setEstimate(cam0);
    cam->setFixed(false);
    graph.addVertex(cam);

    std::vector<g2o::EdgeProjectP2MC *> edges((size_t)n);
    for (int i = 0; i < n; i++) {
        auto *X = new g2o::VertexSBAPointXYZ();
        X->setId(i + 1);
        X->setMarginalized(false);
        X->setEstimate(Eigen::Vector3d(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]));
        X->setFixed(true);
        graph.addVertex(X);
        auto *e = new g2o::EdgeProjectP2MC();
        e->setVertex(0, X);
        e->setVertex(1, cam);
        e->setMeasurement(Eigen::Vector2d((double)obs[2 * i], (double)obs[2 * i + 1]));   // float pixel -> double, as cv::Point2f -> lvt_vector2
        e->information() = Eigen::Matrix2d::Identity();
        auto *rk = new g2o::RobustKernelCauchy;
        e->setRobustKernel(rk);
        rk->setDelta(kernel_width);
        graph.addEdge(e);
        edges[(size_t)i] = e;
    }
    for (int i = 0; i < n; i++) inlier_marks[i] = 1;
    for (int pass = 0; pass < 2; pass++) {
        graph.initializeOptimization(0);
        graph.optimize(5);
        for (int i = 0; i < n; i++)
            if (edges[(size_t)i]->chi2() > th2) {
                edges[(size_t)i]->setLevel(1);
                inlier_marks[i] = 0;
            }
    }
    int inliers = 0;
    for (int i = 0; i < n; i++) {
        inliers += inlier_marks[i];
        if (chi2_out) chi2_out[i] = edges[(size_t)i]->chi2();
    }
    const Eigen::Quaterniond q = cam->estimate().rotation();
    const Eigen::Vector3d t = cam->estimate().translation();
    q_out[0] = q.w(), q_out[1] = q.x(), q_out[2] = q.y(), q_out[3] = q.z();
    p_out[0] = t.x(), p_out[1] = t.y(), p_out[2] = t.z();
    graph.clear();
    return inliers;
}

}  // extern "C"
