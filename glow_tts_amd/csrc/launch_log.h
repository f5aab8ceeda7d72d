// Launch log (diagnostics, always on): host-side counters of kernel launches per kernel class, so that a parity test can assert
// WHICH kernel a call took (e.g. that a full-width backward really went through conv_chain<LINEAR,DGATE> and the LDS-DMA data
// gradient, not through the register-staged fallback).  One mutex-free atomic add per launch; counts launches issued by the host
// (a replayed hipGraph does not count again).  Read with glowtts_launch_count / glowtts_launch_log_dump (include/glowtts_hip.h).
#pragma once
#include <stdio.h>

#include <atomic>

void glowtts_note_launch(const char* cls);
std::atomic<long long>* glowtts_launch_slot(const char* cls);     // never freed; nullptr when the table is full

#define GLOWTTS_NOTE(...) do { char nm_[96]; snprintf(nm_, sizeof(nm_), __VA_ARGS__); glowtts_note_launch(nm_); } while (0)
// the same for call sites whose class name depends only on template parameters: the slot is looked up once per instantiation
#define GLOWTTS_NOTE_STATIC(...) do { static std::atomic<long long>* slot_ = [&] { char nm_[96]; snprintf(nm_, sizeof(nm_), __VA_ARGS__); \
                                          return glowtts_launch_slot(nm_); }(); if (slot_) slot_->fetch_add(1, std::memory_order_relaxed); } while (0)
