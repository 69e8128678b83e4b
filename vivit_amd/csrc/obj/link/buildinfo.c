const char *vivit_hip_source_hash(void) { return "271c69d25bb5d4e83ec99dc0964fa7d6-262e1d34"; }
