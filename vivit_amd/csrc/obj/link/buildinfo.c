const char *vivit_hip_source_hash(void) { return "b06f502518d47d12563a8bb25cd1662d-262e1d34"; }
