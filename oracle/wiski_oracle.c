/*
 * TEST INFRASTRUCTURE -- CPU oracle for the WISKI streaming hot path.
 * See wiski_oracle_impl.h for scope, citations and the "parity unpinned"
 * statement.  Built by oracle/Makefile into oracle/_build/libwiski_oracle.so.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define REAL double
#define SUFFIX _f64
#include "wiski_oracle_impl.h"
#undef REAL
#undef SUFFIX

#define REAL float
#define SUFFIX _f32
#include "wiski_oracle_impl.h"
#undef REAL
#undef SUFFIX
