"""TEST INFRASTRUCTURE ONLY (see rox_oracle.c): CPU restatement of the
reference's trace path + the shim that makes the reference importable."""
