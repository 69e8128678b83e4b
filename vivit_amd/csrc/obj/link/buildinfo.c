const char *vivit_hip_source_hash(void) { return "7ad86f36e7c491df7a58980ab1f96586-262e1d34"; }
