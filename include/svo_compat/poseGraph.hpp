// svo_compat/poseGraph.hpp -- the reference's globalPoseGraph (include/poseGraph.h:36-179) on
// top of the svo_pg_* C ABI.  Same method names, argument meaning and side effects
// (result.g2o written by globalOptimize, poseGraph.g2o by saveStructure); no g2o, no Eigen
// required (see types.hpp for the Isometry3d stand-in / binding).
#pragma once

#include "types.hpp"

namespace svo_compat {

class globalPoseGraph {
  public:
    int globalNodeID = 0;          // include/poseGraph.h:38
    bool loopClosureFlag = false;  // :39
    std::string outFileName = "poseGraph.g2o";  // :56
    int optimizeIterations = 10;   // optimizer.optimize(10), :130
    bool writeResultFile = true;   // optimizer.save("result.g2o"), :131

    explicit globalPoseGraph(svo_ctx *ctx = nullptr) : ctx_(ctx ? ctx : shared_context()) { check(svo_pg_create(ctx_, &pg_)); }
    ~globalPoseGraph() { svo_pg_destroy(pg_); }
    globalPoseGraph(const globalPoseGraph &) = delete;
    globalPoseGraph &operator=(const globalPoseGraph &) = delete;

    // :69-84 -- vertex 0 = identity, fixed
    void initializeGraph()
    {
        check(svo_pg_initialize(pg_));
        globalNodeID = 1;
    }
    // :87-111 -- localT is never read by the reference either
    void augmentNode(const Isometry3d & /*localT*/, const Isometry3d &globalT)
    {
        double p[7];
        iso_to_pose7(globalT, p);
        check(svo_pg_augment_node(pg_, p));
        globalNodeID++;
    }
    // :113-126 -- T is unused by the reference: the measurement is the identity
    void addLoopClosure(const Isometry3d & /*T*/, int fromID)
    {
        check(svo_pg_add_loop_closure(pg_, fromID));
        loopClosureFlag = true;
    }
    // :128-138 -- 10 Gauss-Newton iterations over the whole graph, every estimate returned
    std::vector<Isometry3d> globalOptimize()
    {
        check(svo_pg_optimize(pg_, optimizeIterations, nullptr));
        if (writeResultFile)
            check(svo_pg_write_g2o(pg_, "result.g2o"));
        return estimates();
    }
    std::vector<Isometry3d> estimates() const
    {
        const int n = svo_pg_num_vertices(pg_);
        std::vector<double> p((size_t)n * 7);
        check(svo_pg_get_estimates(pg_, p.data()));
        std::vector<Isometry3d> out;
        out.reserve(n);
        for (int i = 0; i < n; i++)
            out.push_back(pose7_to_iso(&p[7 * i]));
        return out;
    }
    // :140-179
    void saveStructure() { check(svo_pg_write_g2o(pg_, outFileName.c_str())); }
    int numVertices() const { return svo_pg_num_vertices(pg_); }
    int numEdges() const { return svo_pg_num_edges(pg_); }
    svo_posegraph *handle() { return pg_; }

  private:
    svo_ctx *ctx_;
    svo_posegraph *pg_ = nullptr;
};

}  // namespace svo_compat
