// catalogue.hpp — the model catalogue next to the load path: models_directory.json -> ModelInfo, the
// lookup of a model's .bin on disk, and the sha256 the catalogue carries for every file.
//
// Mirrors (same names, same decisions): parse_model_json / load_models_info
// (/root/reference/src/model-utils/model-infos.cpp:114-260), find_model_folder / find_model_bin_file
// (model-downloader.cpp:9-89), find_bin_file_in_folder (model-find-utils.cpp:37-52); ModelInfo / ModelFileDownloadInfo
// (model-downloader-types.h).  The reference downloads the JSON from GitHub first and falls back to the local file; only
// the local file exists here (no network).  The reference PARSES "sha256" and never checks it: verify_model_file is the
// check the field is there for (no counterpart; opt-in through lv_filter_verify_model).  Host-only code.
#pragma once

#include <map>
#include <string>
#include <vector>

enum ModelType { MODEL_TYPE_TRANSCRIPTION = 0, MODEL_TYPE_TRANSLATION = 1 };

struct ModelFileDownloadInfo {
    std::string url;
    std::string sha256;
};

struct ModelInfo {
    std::string friendly_name;
    std::string local_folder_name;
    ModelType type = MODEL_TYPE_TRANSCRIPTION;
    std::vector<ModelFileDownloadInfo> files;
    struct {
        std::string language, description, source;
    } extra;
};

// {"models": [ {...}, ... ]} -> friendly name -> ModelInfo; entries without a string "friendly_name" are skipped with a
// warning, every other field is optional (warnings as in the reference).  Throws std::runtime_error on malformed JSON or
// a missing "models" array.
std::map<std::string, ModelInfo> load_models_info(const std::string &json_text);

// <data_models_dir>/<local_folder_name> if it exists, else <config_models_dir>/<local_folder_name> if it exists, else "";
// "" also for an entry without friendly name, folder name or files (the reference's validity checks)
std::string find_model_folder(const ModelInfo &model_info, const std::string &data_models_dir, const std::string &config_models_dir);
std::string find_bin_file_in_folder(const std::string &model_local_folder_path);  // first *.bin in the folder, "" if none
std::string find_model_bin_file(const ModelInfo &model_info, const std::string &data_models_dir, const std::string &config_models_dir);

// lower-case hex SHA-256 of a file (streamed) / of a byte string; throws std::runtime_error when the file cannot be read
std::string sha256_file(const std::string &path);
std::string sha256_hex(const void *data, size_t n);
// 0: the file's digest equals files[0].sha256 (case-insensitive); 1: it differs; 2: the catalogue carries no digest
int verify_model_file(const ModelInfo &model_info, const std::string &bin_path);
