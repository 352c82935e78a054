// static_shell.cpp -- the drop-in held exactly as rocket_path.cpp holds its problems.
//
// rocket_path.cpp keeps its problems as FILE-SCOPE STATICS (rocket_path.cpp:33-36), registers their addresses in a static table
// (38-44), points g_problemCur at the F3 problem (46), calls init() on every entry from main (65-68), onActivate() on the
// current one (70), forwards keys to it (130-162), and destroys the objects after main has returned.  INTEGRATION.md
// prescribes that form for the batched problems, so this translation unit IS that form, minus the window: the HIP runtime
// is initialised and ~270 MB of HBM are allocated during static initialisation (before main), and rp_batch_destroy -- stream
// synchronise, hipFree, hipStreamDestroy -- runs from static destructors (after main).  headless_shell.cpp holds its problems
// as locals of main and proves nothing about either.
//
//   rp_static [--keys "i n s n s n13 s F4 n s"]
//
// Tokens as in rp_headless: single characters are onKey() (SPACE for ' '), F3 / F4 switch the current problem
// (rocket_path.cpp:148-157), HOME END PGUP PGDN LEFT RIGHT UP DOWN are special keys, nK = K fused steps.  Exit code 0 and
// nothing on stderr is what a healthy run looks like; a box without a GPU prints the constructor's message before main
// starts ("BatchedOneDPathIP: rp_batch_create failed: ...") and main returns 1 -- under GLUT every key would be a no-op.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <string>

#include "batched_problem.h"

#ifndef RP_STATIC_N
#define RP_STATIC_N 1048576      // two 1 Mi-problem batches in static storage
#endif

#ifdef RP_USE_REFERENCE_PROBLEM_H
void repaint() {}      // the shell's own (draw.h:6, rocket_path.cpp:178-182); nothing to post without a window
#endif

static BatchedOneDPathIP g_problem3(RP_STATIC_N, RP_VARIANT_F3, RP_DTYPE_F64);            // was: static OneDPathInteriorPoint g_problem3;
static BatchedOneDPathIP g_problem4(RP_STATIC_N, RP_VARIANT_F4, RP_DTYPE_F32_STATE);      // was: static OneDPath2InteriorPoint g_problem4;

static Problem * g_problems[] =
{
	&g_problem3,
	&g_problem4,
};

static Problem * g_problemCur = &g_problem3;

int main(int argc, char * argv[])
{
	std::string keys = "s";
	for (int i = 1; i < argc; ++i)
	{
		if (!strcmp(argv[i], "--keys") && i + 1 < argc) keys = argv[++i];
		else { fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
	}
	if (!g_problem3.ok() || !g_problem4.ok()) return 1;      // the constructors have said why, before main started

	for (Problem * problem : g_problems)
	{
		problem->init();
	}

	g_problemCur->onActivate();

	std::istringstream in(keys);
	std::string tok;
	while (in >> tok)
	{
		if (tok == "F3" || tok == "F4")
		{
			Problem * p = g_problems[tok == "F3" ? 0 : 1];
			if (p != g_problemCur) { g_problemCur = p; g_problemCur->onActivate(); }
		}
		else if (tok == "SPACE") g_problemCur->onKey(' ');
		else if (tok == "HOME") g_problemCur->onSpecialKey(RP_KEY_HOME);
		else if (tok == "END") g_problemCur->onSpecialKey(RP_KEY_END);
		else if (tok == "PGUP") g_problemCur->onSpecialKey(RP_KEY_PAGE_UP);
		else if (tok == "PGDN") g_problemCur->onSpecialKey(RP_KEY_PAGE_DOWN);
		else if (tok == "LEFT") g_problemCur->onSpecialKey(RP_KEY_LEFT);
		else if (tok == "RIGHT") g_problemCur->onSpecialKey(RP_KEY_RIGHT);
		else if (tok == "UP") g_problemCur->onSpecialKey(RP_KEY_UP);
		else if (tok == "DOWN") g_problemCur->onSpecialKey(RP_KEY_DOWN);
		else if (tok.size() > 1 && tok[0] == 'n') static_cast<BatchedOneDPathIP *>(g_problemCur)->step(atoi(tok.c_str() + 1));
		else if (tok.size() == 1) g_problemCur->onKey((unsigned char)tok[0]);
		else { fprintf(stderr, "unknown key token %s\n", tok.c_str()); return 2; }
	}

	return 0;      // ~BatchedOneDPathIP of both statics runs after this
}
