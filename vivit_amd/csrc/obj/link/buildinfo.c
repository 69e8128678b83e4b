const char *vivit_hip_source_hash(void) { return "3d3f5573b26968bead646a9443d6df3e-262e1d34"; }
