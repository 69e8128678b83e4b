const char *vivit_hip_source_hash(void) { return "d4400c529f912db1d03f276a7cb0dbb5-262e1d34"; }
