#include "svo_oracle.h"
