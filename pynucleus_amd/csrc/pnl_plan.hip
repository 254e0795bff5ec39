#include "pnl_context.h"
