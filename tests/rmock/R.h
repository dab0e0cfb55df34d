/* tests/rmock/R.h -- see Rinternals.h beside it: a test stand-in, not R. */
#ifndef RMOCK_R_H
#define RMOCK_R_H
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#endif
