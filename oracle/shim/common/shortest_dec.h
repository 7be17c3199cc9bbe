/* shim: see ../postgres.h */
#ifndef PGV_SHIM_SHORTEST_DEC_H
#define PGV_SHIM_SHORTEST_DEC_H
#define FLOAT_SHORTEST_DECIMAL_LEN 16
int			float_to_shortest_decimal_buf(float f, char *result);
#endif
