/* TEST INFRASTRUCTURE ONLY - see oracle_model.c */
#ifndef ORACLE_MODEL_H_
#define ORACLE_MODEL_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
int orc_model_version (void);
#ifdef __cplusplus
}
#endif
#endif
