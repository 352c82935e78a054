// problem_iface.h -- the plug-in interface rocket_path.cpp dispatches through.
//
// The reference declares it in problem.h:3-14: eight pure virtuals, all returning void, no
// error channel.  A batched problem must derive from that very type to sit in
// rocket_path.cpp's g_problems[] table (rocket_path.cpp:38-44), so when this header is
// compiled inside the reference tree define RP_USE_REFERENCE_PROBLEM_H and the reference's
// own problem.h is used; stand-alone builds (this repo, the headless shell) get the
// equivalent declaration below.
#pragma once

#ifdef RP_USE_REFERENCE_PROBLEM_H
#include "problem.h"
#else
struct Problem {
    virtual ~Problem() {}
    virtual void init() = 0;                    // called once for every registered problem at start-up
    virtual void onActivate() = 0;              // problem becomes current (F-key switch): print the help text
    virtual void onKey(unsigned char key) = 0;  // 'n' = one Newton step, 'i'/'j' = re-init, 's' = print, ' ' = feasibility move
    virtual void onSpecialKey(int key) = 0;     // GLUT special keys: nudge variables
    virtual void onDraw() = 0;
    virtual void onMouseMove(int x, int y) = 0;
    virtual void onMouseDown() = 0;
    virtual void onMouseUp() = 0;
};
#endif

// GLUT special-key codes (the public GLUT API values) so the stand-alone build needs no GL headers.
enum {
    RP_KEY_F1 = 1, RP_KEY_F2 = 2, RP_KEY_F3 = 3, RP_KEY_F4 = 4,
    RP_KEY_LEFT = 100, RP_KEY_UP = 101, RP_KEY_RIGHT = 102, RP_KEY_DOWN = 103,
    RP_KEY_PAGE_UP = 104, RP_KEY_PAGE_DOWN = 105, RP_KEY_HOME = 106, RP_KEY_END = 107
};
