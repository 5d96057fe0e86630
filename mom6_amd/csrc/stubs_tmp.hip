#include "mom6x_dev.h"
void bt_state_free(mom6x_ctx *) {}
void rk2_state_free(mom6x_ctx *) {}
