// Minimal stand-in for <pluginlib/class_list_macros.h> (TEST ONLY): the export macro expands to nothing.
#pragma once
#define PLUGINLIB_EXPORT_CLASS(cls, base)
