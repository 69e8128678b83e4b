const char *vivit_hip_source_hash(void) { return "6b57c65382a235c4bab3fd20043479bb-262e1d34"; }
