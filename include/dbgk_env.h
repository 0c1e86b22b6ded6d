/* dbgk_env.h -- how the libraries read their environment (libdbgk.so and the C++ host layer share this header).
 *
 * Three classes, so that the product does not carry switches that change what it computes:
 *   1. CONFIGURATION: plain getenv() of the variables documented in INTEGRATION.md ("Environment"): device choice, batch and
 *      store sizes, thread counts, dumps, timings.  They never change a result.
 *   2. TEST HOOKS: DBGK_TEST_HOOKS="name=value,name=value" -- dbgk_hook("name").  They force a code path the library would
 *      otherwise choose by itself (a kernel form, the exact build, a copy strategy) so that the test-suite can reach every
 *      path on small inputs; every path gives the same result.  Read at each use: the tests change them between handles.
 *   3. EXPERIMENTS: DBGK_EXPERIMENT_ENV("DBGK_...") -- timing switches of profiles/ (tile geometry, phase cut-offs that leave
 *      WRONG results behind, overlap schedules).  They exist only in a build with -DDBGK_EXPERIMENTS
 *      (profiles/tools/build_variant.sh <name> <src> -DDBGK_EXPERIMENTS); in the product build the macro is a null pointer and
 *      the code behind it is dead.
 */
#ifndef DBGK_ENV_H
#define DBGK_ENV_H

#include <stdlib.h>
#include <string.h>

#ifdef DBGK_EXPERIMENTS
#define DBGK_EXPERIMENT_ENV(name) getenv(name)
#else
#define DBGK_EXPERIMENT_ENV(name) ((const char *)0)
#endif

/* value of `name` in DBGK_TEST_HOOKS, or a null pointer; the returned string lives until the thread's next call */
static inline const char *dbgk_hook(const char *name)
{
	const char *e = getenv("DBGK_TEST_HOOKS");
	if (!e) return 0;
	static __thread char val[64];
	const size_t n = strlen(name);
	for (const char *p = e; *p;) {
		const char *q = strchr(p, ',');
		const size_t len = q ? (size_t)(q - p) : strlen(p);
		if (len > n && p[n] == '=' && !strncmp(p, name, n)) {
			size_t m = len - n - 1;
			if (m > sizeof(val) - 1) m = sizeof(val) - 1;
			memcpy(val, p + n + 1, m);
			val[m] = 0;
			return val;
		}
		if (!q) break;
		p = q + 1;
	}
	return 0;
}

#endif
