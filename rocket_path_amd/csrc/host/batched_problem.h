// batched_problem.h -- the drop-in: N interior-point problems behind the reference's Problem.
//
// BatchedOneDPathIP replaces OneDPathInteriorPoint (onedpath_ip.h:5-17, F3) or
// OneDPath2InteriorPoint (onedpath2_ip.h:5-17, F4) in rocket_path.cpp's table
// (rocket_path.cpp:33-44): same virtuals, same key meanings, but every key acts on all N
// problems of a batch that lives in HBM, through the C ABI of include/rp_batch.h.
// With n = 1 and the default init it is the reference's single-problem behaviour.
#pragma once

#include <cstddef>
#include <cstdint>
#include <vector>

#include "problem_iface.h"
#include "rp_batch.h"

struct BatchedOneDPathIP : public Problem {
    // variant: RP_VARIANT_F3 / RP_VARIANT_F4; dtype: RP_DTYPE_F64 / RP_DTYPE_F32
    explicit BatchedOneDPathIP(size_t n = 1, int variant = RP_VARIANT_F3, int dtype = RP_DTYPE_F64, int device = 0);
    ~BatchedOneDPathIP() override;

    // ---- the reference interface (problem.h:3-14) ----
    void init() override;                  // initDefault on every problem (onedpath_ip.cpp:230-233)
    void onActivate() override;            // help text (onedpath_ip.cpp:235-248)
    void onKey(unsigned char key) override;    // onedpath_ip.cpp:250-278
    void onSpecialKey(int key) override;       // onedpath_ip.cpp:280-324
    void onDraw() override;                // fills plotPositions()/plotAccelerations() for the watched problem (70 doubles read back);
                                           // the GL calls that draw them (onedpath_ip.cpp:336-358, 1015-1148) are the shell's to add
    void onMouseMove(int, int) override {}
    void onMouseDown() override {}
    void onMouseUp() override {}

    // ---- batch extras (no counterpart in the reference) ----
    bool ok() const { return batch_ != nullptr; }   // false: no GPU / creation failed (message printed)
    size_t size() const { return n_; }
    void setProblems(const double *pos0, const double *pos1, const double *pos2);   // feasible-start rule
    void step(int k);                               // k x onKey('n') in one launch
    void solve(double gapTol = 1e-8, int maxIter = 200);
    void watch(size_t index) { watched_ = index < n_ ? index : 0; }   // which problem 's' prints / onDraw plots
    bool readState(std::vector<double> &aos);       // n * 16 (F3) or n * 12 (F4) doubles
    bool readIters(std::vector<int32_t> &iters, std::vector<uint32_t> &status);
    bool reduce(rp_reduction &out);
    const std::vector<double> &plotPositions() const { return plotPos_; }
    const std::vector<double> &plotAccelerations() const { return plotAcc_; }
    rp_batch *handle() { return batch_; }
    // Stand-alone hosts: called after every handled key, where the reference calls repaint() (draw.h:6).  Built inside the
    // reference tree (RP_USE_REFERENCE_PROBLEM_H) the class calls the shell's own repaint() and this hook is unused.
    void setRepaintHook(void (*hook)()) { repaintHook_ = hook; }

private:
    void printState();
    void requestRepaint();
    bool check(int status, const char *what);
    rp_batch *batch_;
    size_t n_, watched_;
    int variant_;
    std::vector<double> plotPos_, plotAcc_;
    void (*repaintHook_)();
};
