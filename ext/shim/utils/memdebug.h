#include "pgshim_ref.h"
