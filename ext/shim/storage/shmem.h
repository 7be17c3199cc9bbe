#include "pgshim.h"
