/* shim: see postgres.h in this directory */
