/* placeholder until the SGNS restatement lands (keeps the Makefile stable) */
#include <stdint.h>
int64_t orc_sgns_train(void) { return -1; }
