"""Stand-in namespace (see ../README.md). Pure torch; our own code."""
