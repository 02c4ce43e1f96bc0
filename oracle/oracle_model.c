/*
 * TEST INFRASTRUCTURE ONLY.  Plain-C restatement of the recompressor's context-index
 * computation.  Filled in together with the ctx-index kernel.
 */
#include "oracle_model.h"
int orc_model_version (void) { return 0; }
