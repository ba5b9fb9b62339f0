// empty on purpose: see ../host_shim.h (oracle/ref_shim, test infrastructure)
